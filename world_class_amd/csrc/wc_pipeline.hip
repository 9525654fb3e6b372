// Fused analysis + synthesis pipeline (extension; the reference has no such entry point -- its demo calls the
// four stages one after the other, reference test/test.cpp:288-384).  One call enqueues the whole hot path
// for a packed batch without any host synchronisation in between.  CheapTrick, D4C and the Synthesis time base
// only depend on the F0 contour; the noise-stream positions are chained on the device (CheapTrick end -> D4C
// start -> Synthesis start), exactly the order a single reference process would consume its global randn()
// stream.  Capacity overflows of the rate-bounded buffers (Harvest zero crossings, Synthesis pulses) are
// checked once at the end and re-run with hard bounds.
//
// Default schedule ("groups"): the batch is cut into two halves A and B, each with its own stage handles and
// two streams (main: Harvest, time base, pulses; aux: CheapTrick frames, D4C), and events order the ALU-bound
// kernels one after the other while every latency-bound stretch runs in the shadow of an ALU-bound one:
//
//   main A  Harvest_A heavy | tail_A ........ | time base_A ....................... | pulses_A
//   main B                  | Harvest_B heavy | tail_B ............ | time base_B ............... | pulses_B
//   aux A                                     | CheapTrick_A, D4C_A ............... |
//   aux B                                                                           | CheapTrick_B, D4C_B |
//
// (heavy = band-pass, raw candidates, refinement; tail = unreliable-candidate test, contour logic, smoothing).
// Measured on MI355X: overlapping two ALU-bound kernels (refinement next to D4C) gains nothing -- they time-share
// the CUs -- while the tails and time bases (64 wavefronts each) disappear behind the heavy kernels.
// WC_PIPELINE_MODE=shared selects the older schedule: one set of stages, Harvest split over two streams.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "wc_stages.hpp"
#include "wc_hostcopy.hpp"
#include "../../include/world_class_io.h"
#include "../../include/world_class_codec.h"

using namespace wc;

// one independent chain (mode "groups"): its own stage handles and two streams
constexpr int kMaxGroups = 8;  // two for a device-resident batch; up to eight (of growing size) when the rows leave for the host
constexpr double kUnchainBelowSeconds = 500.0;  // (see pipeline_run: WC_PIPELINE_UNCHAIN_BELOW)
struct PipeGroup {
	wc_harvest *hv = nullptr;
	wc_cheaptrick *ct = nullptr;
	wc_d4c *d4 = nullptr;
	wc_synthesis *sy = nullptr;
	hipStream_t main = nullptr, aux = nullptr;
	hipStream_t aux_hi = nullptr;  // group 0: the pipeline's high-priority stream s1_hi (runs whose rows leave for the host)
	hipEvent_t e0 = nullptr, e_aux = nullptr, e_mid = nullptr, e_ct = nullptr;
};

// Host-side copies of the batch front-end, spread over a few threads: one thread moves 5-10 GB/s, the 2.1 GB of
// spectrogram + aperiodicity rows of a 64 x 10 s batch would take a quarter of a second on one.
// WC_PIPELINE_TIMING=1: wall-clock marks of the host front-end's phases on stderr (development aid)
static std::chrono::steady_clock::time_point g_mark0;
static void pmark(const char *what) {
	static const bool on = getenv("WC_PIPELINE_TIMING") != nullptr;
	if (on) std::fprintf(stderr, "  [pipeline] %-30s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - g_mark0).count());
}
// What wc_pipeline_run_batch_host asks of a run: the spectrogram / aperiodicity rows of each half batch start their way
// to the host as soon as that half's D4C is through (copy stream, pinned staging), and are handed to the caller's
// per-utterance buffers while the other half still computes.
struct HostSink {
	char *stage_sp = nullptr, *stage_ap = nullptr;  // pinned, packed [frames][bins] like the device arrays; nullptr = not wanted
	double *const *sp = nullptr, *const *ap = nullptr;  // the caller's per-utterance destinations
	const int *f_len = nullptr;
	int bins = 0;
	bool overlapped[kMaxGroups] = {};  // group g was copied and scattered during the run (first attempt only)
	bool direct = false;  // every destination row lies in pinned host memory: the copy engine writes the rows where they belong
	// the waveforms (float64) straight into the caller's page-locked rows, each half as soon as its pulses are summed
	double *const *y = nullptr;
	const int *y_len = nullptr;
	bool y_done = false;
	hipEvent_t x_ev[kMaxGroups] = {};  // the samples of group g > 0 are on the device (their upload runs beside the first group's Harvest)
	bool eager = false;        // a group's CheapTrick / D4C do not wait for the next group's Harvest: their rows leave earlier
	int ng = 0;                // > 0: the run is cut into ng groups, group g = utterances [ub[g], ub[g + 1]) (a run whose rows leave for
	int ub[kMaxGroups + 1] = {};  // the host takes SMALL first groups: below)
};

// true when p lies in page-locked host memory (hipHostMalloc / hipHostRegister; e.g. a pinned torch tensor)
static bool is_pinned(const void *p) {
	hipPointerAttribute_t at;
	if (hipPointerGetAttributes(&at, p) != hipSuccess) {
		(void)hipGetLastError();
		return false;
	}
	return at.type == hipMemoryTypeHost;
}

// Schedule knobs (development / A-B switches).  Read from the environment ONCE, when the handle is created, and changed afterwards
// through wc_pipeline_set_option only: a run reads no environment variable (round-5 advice: getenv beside another thread's
// setenv is a data race, and a run made several such reads per attempt).
struct PipeKnobs {
	double unchain_below = kUnchainBelowSeconds;  // WC_PIPELINE_UNCHAIN_BELOW / "unchain_below": seconds of signal below which the two groups run side by side
	int schedule_chains = 0;   // WC_PIPELINE_SCHEDULE=chains / "schedule": the four-stream order instead of two lanes
	int side = 0;              // WC_PIPELINE_SIDE=h / c ('h', 'c') / "side"
	int syn_streams = 1;       // WC_PIPELINE_SYN_STREAMS / "syn_streams"
	int chain_min = 4;         // WC_PIPELINE_CHAIN_MIN / "chain_min"
	int groups = 2;            // WC_PIPELINE_GROUPS / "groups"
	int tail_after_bp = 0;     // WC_PIPELINE_TAIL_AFTER_BP / "tail_after_bp"
	int chain = 1;             // WC_PIPELINE_CHAIN / "chain"
	int direct = 1;            // WC_PIPELINE_DIRECT / "direct"
	int eager = 1;             // WC_PIPELINE_EAGER / "eager"
	int pre_lane = 0;          // WC_PIPELINE_PRE_LANE / "pre_lane" (experiment, measured and not kept): the second group's decimation .. seam values on the latency lane (1) or a stream of their own (2)
	int force_tie = -1;        // WC_PIPELINE_FORCE_TIE / "force_tie" (test hook): this utterance of every run counts as flagged for a tie
	std::vector<int> host_splits;  // WC_PIPELINE_HOST_SPLITS "a,b,c" / WC_PIPELINE_HOST_SPLIT "a" / "host_splits": per cent of the utterances per group but the last
	bool host_splits_set = false;
};

static void knob_set(PipeKnobs &k, const std::string &name, const char *v) {
	const std::string val = v ? v : "";
	if (name == "unchain_below") k.unchain_below = v ? atof(v) : kUnchainBelowSeconds;
	else if (name == "schedule") k.schedule_chains = val == "chains" ? 1 : 0;
	else if (name == "side") k.side = val.empty() ? 0 : val[0];
	else if (name == "syn_streams") k.syn_streams = val.empty() ? 1 : atoi(v);
	else if (name == "chain_min") k.chain_min = val.empty() ? 4 : atoi(v);
	else if (name == "groups") k.groups = val.empty() ? 2 : atoi(v);
	else if (name == "tail_after_bp") k.tail_after_bp = val.empty() ? 0 : atoi(v);
	else if (name == "chain") k.chain = val.empty() ? 1 : atoi(v);
	else if (name == "direct") k.direct = val.empty() ? 1 : atoi(v);
	else if (name == "eager") k.eager = val.empty() ? 1 : atoi(v);
	else if (name == "force_tie") k.force_tie = val.empty() ? -1 : atoi(v);
	else if (name == "pre_lane") k.pre_lane = val.empty() ? 0 : atoi(v);
	else if (name == "host_splits") {
		k.host_splits.clear();
		k.host_splits_set = !val.empty();
		for (const char *c = val.c_str(); *c && (int)k.host_splits.size() < kMaxGroups - 1;) {
			k.host_splits.push_back(atoi(c));
			while (*c && *c != ',') ++c;
			if (*c == ',') ++c;
		}
	}
}

struct wc_pipeline {
	int mode;  // 0: shared stages, Harvest split over streams; 1: independent staggered chains per utterance group
	PipeKnobs *knobs;
	PipeGroup grp[kMaxGroups];
	int n_grp;  // groups whose handles exist (2 at creation, the others when a run first asks for them)
	double c_floor, c_ceil, c_q1, c_ct_floor, c_threshold;  // what the stage handles of a group are created with
	int c_fft_size;
	bool copy_prio;
	int prio_hi;
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
	hipStream_t s1, s1_hi, s2, s_copy, s_copy2;  // (s_copy2: the aperiodicity rows leave beside the spectrogram rows, on a DMA engine of their own)
	hipEvent_t e_copy2[kMaxGroups];
	hipEvent_t e0, e1, e2, e_copy[kMaxGroups], e_y[kMaxGroups], e_ycopy[kMaxGroups], e_x[kMaxGroups], e_bp;
	hipEvent_t tm0, tm[kMaxGroups][4];  // WC_PIPELINE_TIMING: time-stamped marks of every group's progress (created on first use)
	// host batch front-end (wc_pipeline_run_batch_host): device-resident batch + pinned staging, grow-only
	DevBuf b_x, b_pcm, b_t, b_f, b_sp, b_ap, b_y, b_ypcm, b_coded;
	HostBuf st_in, st_out;
};

// the stage handles, streams and events of groups [p->n_grp, ng)
static int pipeline_ensure_groups(wc_pipeline *p, int ng) {
	OnDeviceOf here(p->dev);  // (groups beyond the second are created by the first run that asks for them, on whichever thread that is)
	for (int g = p->n_grp; g < ng && g < kMaxGroups; ++g) {
		PipeGroup &G = p->grp[g];
		G.hv = wc_harvest_create(p->fs, p->c_floor, p->c_ceil, p->frame_period, 8000.0, 40.0, 0);
		G.ct = G.hv ? wc_cheaptrick_create(p->fs, p->c_q1, p->c_ct_floor, p->c_fft_size) : nullptr;
		G.d4 = G.ct ? wc_d4c_create(p->fs, p->c_threshold) : nullptr;
		G.sy = G.d4 ? wc_synthesis_create(p->fs, p->fft_size, p->frame_period) : nullptr;
		if (!G.sy) return WC_ERR_DEVICE;  // (the failed create has set the error text; wc_pipeline_destroy releases what exists)
		// the contour these stages see comes out of Harvest: candidates outside [floor, ceil] are struck (reference
		// src/harvest.cpp:974-979) and the smoothing filter overshoots by a few per cent at most
		ct_set_f0_bound(G.ct, 1.25 * p->c_ceil);
		d4c_set_f0_bound(G.d4, 1.25 * p->c_ceil);
		if (g == 0) { G.main = p->dev->active(); G.aux = p->s1; G.aux_hi = p->s1_hi; }
		else if (g == 1) { G.main = p->n_split > 1 ? p->hs[1] : p->s2; G.aux = p->s2; }
		// (groups 2 .. 5 have no streams of their own: HIP multiplexes the streams of one priority onto four hardware queues, and a
		// stream that shares a queue with one waiting for an event stands still with it -- measured: four groups on streams of
		// their own 72 ms, with GPU_MAX_HW_QUEUES=8 54 ms.  They take turns on the two main streams and share the one
		// high-priority stream, see pipeline_run.)
		WC_HIP(hipEventCreateWithFlags(&G.e0, hipEventDisableTiming));
		WC_HIP(hipEventCreateWithFlags(&G.e_aux, hipEventDisableTiming));
		WC_HIP(hipEventCreateWithFlags(&G.e_mid, hipEventDisableTiming));
		WC_HIP(hipEventCreateWithFlags(&G.e_ct, hipEventDisableTiming));
		p->n_grp = g + 1;
	}
	return WC_OK;
}

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
	p->c_floor = harvest_f0_floor; p->c_ceil = harvest_f0_ceil; p->c_q1 = q1; p->c_ct_floor = cheaptrick_f0_floor;
	p->c_threshold = d4c_threshold; p->c_fft_size = fft_size;
	p->knobs = new PipeKnobs();
	for (const char *nm : {"UNCHAIN_BELOW", "SCHEDULE", "SIDE", "SYN_STREAMS", "CHAIN_MIN", "GROUPS", "TAIL_AFTER_BP", "CHAIN", "DIRECT", "EAGER", "FORCE_TIE", "HOST_SPLITS", "PRE_LANE"}) {
		const std::string env = std::string("WC_PIPELINE_") + nm;
		std::string low(nm);
		for (char &c : low) c = (char)tolower(c);
		if (const char *v = getenv(env.c_str())) knob_set(*p->knobs, low, v);
	}
	if (!p->knobs->host_splits_set)
		if (const char *v = getenv("WC_PIPELINE_HOST_SPLIT")) knob_set(*p->knobs, "host_splits", v);
	{
		const char *m = getenv("WC_PIPELINE_MODE");
		p->mode = (m && std::string(m) == "shared") ? 0 : 1;  // default: two scheduled chains (measured 82 vs 90-96 ms per batch)
	}
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
	if (ok) {  // (the contour these stages see comes out of Harvest, see below)
		ct_set_f0_bound(p->ct, 1.25 * harvest_f0_ceil);
		d4c_set_f0_bound(p->d4, 1.25 * harvest_f0_ceil);
	}
	ok = ok && hipStreamCreateWithFlags(&p->s1, hipStreamNonBlocking) == hipSuccess;
	ok = ok && hipStreamCreateWithFlags(&p->s2, hipStreamNonBlocking) == hipSuccess;
	ok = ok && hipEventCreateWithFlags(&p->e0, hipEventDisableTiming) == hipSuccess;
	ok = ok && hipEventCreateWithFlags(&p->e1, hipEventDisableTiming) == hipSuccess;
	ok = ok && hipEventCreateWithFlags(&p->e2, hipEventDisableTiming) == hipSuccess;
	// The copy streams are created with a priority of their own: HIP multiplexes the streams of one priority onto a few hardware
	// queues (four by default), and a copy that shares a queue with a compute stream sits behind that stream's kernels -- and
	// holds up the kernels behind it while it waits for its rows (measured: the second half's CheapTrick started 15 ms late
	// behind the first half's rows).  Streams of another priority get queues of their own.  WC_PIPELINE_COPY_PRIORITY=0: plain streams (A/B).
	int prio_lo = 0, prio_hi = 0;
	(void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
	const bool copy_prio = !(getenv("WC_PIPELINE_COPY_PRIORITY") && getenv("WC_PIPELINE_COPY_PRIORITY")[0] == '0') && prio_hi != prio_lo;
	p->copy_prio = copy_prio;
	p->prio_hi = prio_hi;
	auto copy_stream = [&](hipStream_t *st) {
		return (copy_prio ? hipStreamCreateWithPriority(st, hipStreamNonBlocking, prio_hi) : hipStreamCreateWithFlags(st, hipStreamNonBlocking)) == hipSuccess;
	};
	ok = ok && copy_stream(&p->s_copy);
	// ... and so does the stream the first half's CheapTrick / D4C take in a run whose rows leave for the host: there they are
	// enqueued beside the second half's Harvest (HostSink::eager) and should get the CUs first, because PCIe waits for their rows
	// (measured: 61 -> 58 ms per 64 utterances with all five outputs).  WC_PIPELINE_AUX_PRIORITY=0: the plain stream.
	if (ok && copy_prio && !(getenv("WC_PIPELINE_AUX_PRIORITY") && getenv("WC_PIPELINE_AUX_PRIORITY")[0] == '0'))
		ok = hipStreamCreateWithPriority(&p->s1_hi, hipStreamNonBlocking, prio_hi) == hipSuccess;
	{
		const char *env = getenv("WC_PIPELINE_COPY_STREAMS");
		if (ok && !(env && atoi(env) == 1)) {
			ok = copy_stream(&p->s_copy2);
			for (int g = 0; g < kMaxGroups; ++g) ok = ok && hipEventCreateWithFlags(&p->e_copy2[g], hipEventDisableTiming) == hipSuccess;
		}
	}
	for (int g = 0; g < kMaxGroups; ++g) {
		ok = ok && hipEventCreateWithFlags(&p->e_copy[g], hipEventDisableTiming) == hipSuccess;
		ok = ok && hipEventCreateWithFlags(&p->e_y[g], hipEventDisableTiming) == hipSuccess;
		ok = ok && hipEventCreateWithFlags(&p->e_ycopy[g], hipEventDisableTiming) == hipSuccess;
		ok = ok && hipEventCreateWithFlags(&p->e_x[g], hipEventDisableTiming) == hipSuccess;
	}
	ok = ok && hipEventCreateWithFlags(&p->e_bp, hipEventDisableTiming) == hipSuccess;
	if (ok && p->mode == 1) ok = pipeline_ensure_groups(p, 2) == WC_OK;
	// (the handle that re-runs utterances on a tie, created here rather than inside the first run that meets one: building its tables
	// is a few milliseconds of host work and synchronous uploads; its workspaces grow when it is first used)
	if (ok) ok = hv_exact_twin(p->hv[0]) != nullptr;
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
	if (p->dev) p->dev->quiesce();
	for (DevBuf *b : {&p->b_x, &p->b_pcm, &p->b_t, &p->b_f, &p->b_sp, &p->b_ap, &p->b_y, &p->b_ypcm, &p->b_coded}) b->release();
	p->st_in.release();
	p->st_out.release();
	if (p->s1) { (void)hipStreamSynchronize(p->s1); (void)hipStreamDestroy(p->s1); }
	if (p->s2) { (void)hipStreamSynchronize(p->s2); (void)hipStreamDestroy(p->s2); }
	if (p->s1_hi) { (void)hipStreamSynchronize(p->s1_hi); (void)hipStreamDestroy(p->s1_hi); }
	if (p->e0) (void)hipEventDestroy(p->e0);
	if (p->e1) (void)hipEventDestroy(p->e1);
	if (p->e2) (void)hipEventDestroy(p->e2);
	if (p->s_copy) (void)hipStreamDestroy(p->s_copy);
	if (p->s_copy2) (void)hipStreamDestroy(p->s_copy2);
	for (int g = 0; g < kMaxGroups; ++g) if (p->e_copy2[g]) (void)hipEventDestroy(p->e_copy2[g]);
	for (int g = 0; g < kMaxGroups; ++g) if (p->e_copy[g]) (void)hipEventDestroy(p->e_copy[g]);
	for (int g = 0; g < kMaxGroups; ++g) if (p->e_y[g]) (void)hipEventDestroy(p->e_y[g]);
	for (int g = 0; g < kMaxGroups; ++g) if (p->e_ycopy[g]) (void)hipEventDestroy(p->e_ycopy[g]);
	for (int g = 0; g < kMaxGroups; ++g) if (p->e_x[g]) (void)hipEventDestroy(p->e_x[g]);
	if (p->e_bp) (void)hipEventDestroy(p->e_bp);
	if (p->tm0) {
		(void)hipEventDestroy(p->tm0);
		for (int g = 0; g < kMaxGroups; ++g) for (int k = 0; k < 4; ++k) if (p->tm[g][k]) (void)hipEventDestroy(p->tm[g][k]);
	}
	for (int g = 0; g < kMaxGroups; ++g) {
		PipeGroup &G = p->grp[g];
		if (G.e0) (void)hipEventDestroy(G.e0);
		if (G.e_aux) (void)hipEventDestroy(G.e_aux);
		if (G.e_mid) (void)hipEventDestroy(G.e_mid);
			if (G.e_ct) (void)hipEventDestroy(G.e_ct);
		wc_synthesis_destroy(G.sy);
		wc_d4c_destroy(G.d4);
		wc_cheaptrick_destroy(G.ct);
		wc_harvest_destroy(G.hv);
	}
	wc_synthesis_destroy(p->sy);
	wc_d4c_destroy(p->d4);
	wc_cheaptrick_destroy(p->ct);
	for (int k = 0; k < 4; ++k) {
		if (p->hs[k]) { (void)hipStreamSynchronize(p->hs[k]); (void)hipStreamDestroy(p->hs[k]); }
		if (p->he[k]) (void)hipEventDestroy(p->he[k]);
		wc_harvest_destroy(p->hv[k]);
	}
	delete p->knobs;
	delete p;
}

// A schedule knob of this handle (see PipeKnobs; value NULL or "" = the default): "unchain_below", "schedule", "side", "syn_streams",
// "chain_min", "groups", "tail_after_bp", "chain", "direct", "eager", "host_splits", "force_tie".  Not for use beside a run of the same handle.
int wc_pipeline_set_option(wc_pipeline *p, const char *name, const char *value) {
	if (!p || !name) return fail(WC_ERR_INVALID, "pipeline option: null argument");
	static const char *known[] = {"unchain_below", "schedule", "side", "syn_streams", "chain_min", "groups", "tail_after_bp", "chain", "direct", "eager",
								  "host_splits", "force_tie", "pre_lane"};
	bool ok = false;
	for (const char *k : known) ok = ok || std::strcmp(k, name) == 0;
	if (!ok) return fail(WC_ERR_INVALID, "pipeline option: unknown name");
	DeviceLock lock(p->dev);
	knob_set(*p->knobs, name, value);
	return WC_OK;
}

int wc_pipeline_get_fft_size(const wc_pipeline *p) { return p ? p->fft_size : WC_ERR_INVALID; }

static int pipeline_run(wc_pipeline *p, int n_utt, const double *d_x, const int *x_length, double *d_tpos, double *d_f0,
						double *d_sp, double *d_ap, double *d_y, uint64_t *rng_pos, HostSink *sink) {
	if (!p || n_utt <= 0 || !d_x || !x_length || !d_tpos || !d_f0 || !d_sp || !d_ap || !d_y)
		return fail(WC_ERR_INVALID, "pipeline: null argument");
	WC_HIP(hipSetDevice(p->dev->id));
	DeviceLock lock(p->dev);
	Device *dev = p->dev;
	hipStream_t s0 = dev->active();
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
	// the start positions are read from a private copy: rng_pos is also the output and a capacity retry re-reads them
	std::vector<uint64_t> rng_in;
	if (rng_pos) rng_in.assign(rng_pos, rng_pos + n_utt);
	const uint64_t *rng_start = rng_pos ? rng_in.data() : nullptr;
	// Round 6: utterances in whose refinement a raw candidate sat on a tie of the window length or a harmonic's bin (hv_refine_packed_kernel)
	// go through the path once more BY THEMSELVES, the band-pass as direct FIR sums: stretch of consecutive utterances by stretch,
	// on the caller's stream and the handle's own plain stage handles, straight into their slices of the outputs -- everything the
	// other utterances produced stands (an utterance's noise positions are its own).  Until round 5 one flagged utterance sent
	// every stage of every group through a second attempt.
	auto fix_up = [&](std::vector<int> tied) -> int {
		if (p->knobs->force_tie >= 0 && p->knobs->force_tie < n_utt && tied.empty()) tied.push_back(p->knobs->force_tie);  // (test hook)
		if (tied.empty()) return WC_OK;
		wc_harvest *t = hv_exact_twin(p->hv[0]);
		if (!t) return WC_ERR_DEVICE;
		std::vector<long long> xo(n_utt + 1, 0), fo(n_utt + 1, 0), yo(n_utt + 1, 0);
		for (int u = 0; u < n_utt; ++u) { xo[u + 1] = xo[u] + x_length[u]; fo[u + 1] = fo[u] + f_len[u]; yo[u + 1] = yo[u] + y_len[u]; }
		for (const auto &r : hv_runs_of(tied)) {
			const int u0 = r.first, nu = r.second - r.first;
			const double *gx = d_x + xo[u0];
			double *gt = d_tpos + fo[u0], *gf = d_f0 + fo[u0], *gsp = d_sp + fo[u0] * bins, *gap = d_ap + fo[u0] * bins, *gy = d_y + yo[u0];
			bool hv_full = false, syn_full = false, done = false;
			for (int attempt = 0; attempt < 3 && !done; ++attempt) {
				int rc2;
				long long total = 0;
				uint64_t a0 = 0, a1 = 0;
				if ((rc2 = hv_enqueue(t, s0, nu, gx, x_length + u0, gt, gf, hv_full, nullptr, nullptr))) return rc2;
				if ((rc2 = ct_prepare(p->ct, s0, nu, x_length + u0, gf, f_len.data() + u0, rng_start ? rng_start + u0 : nullptr, &total, &a0, &a1))) return rc2;
				if ((rc2 = ct_frames(p->ct, s0, nu, gx, gt, gf, gsp, total, nullptr))) return rc2;
				if ((rc2 = d4c_enqueue(p->d4, s0, nu, gx, x_length + u0, gt, gf, f_len.data() + u0, p->fft_size, gap, nullptr, ct_end_positions(p->ct)))) return rc2;
				if ((rc2 = syn_prepare(p->sy, s0, nu, gf, f_len.data() + u0, y_len.data() + u0, gy, nullptr, syn_full))) return rc2;
				if ((rc2 = syn_pulses(p->sy, s0, gf, gsp, gap, gy, d4c_end_positions(p->d4)))) return rc2;
				bool o1 = false, o2 = false;
				if ((rc2 = syn_finish(p->sy, s0, rng_pos ? rng_pos + u0 : nullptr, &o2))) return rc2;
				if ((rc2 = hv_overflowed(t, s0, &o1, nullptr))) return rc2;
				hv_full = hv_full || o1;
				syn_full = syn_full || o2;
				done = !o1 && !o2;
			}
			if (!done) return fail(WC_ERR_DEVICE, "pipeline: buffer overflow");
			if (sink) {  // rows (and waveforms) that left for the host during the run: these utterances' once more
				std::vector<CopyJob> jobs;
				for (int u = u0; u < u0 + nu; ++u) {
					const size_t ulen = sizeof(double) * (size_t)f_len[u] * bins, off = sizeof(double) * (size_t)fo[u] * bins;
					for (int which = 0; which < 2; ++which) {
						char *stage = which == 0 ? sink->stage_sp : sink->stage_ap;
						double *const *rows = which == 0 ? sink->sp : sink->ap;
						const double *src = (which == 0 ? d_sp : d_ap) + fo[u] * bins;
						if (!stage || !rows || !rows[u]) continue;
						if (sink->direct) WC_HIP(hipMemcpyAsync(rows[u], src, ulen, hipMemcpyDeviceToHost, s0));
						else {
							WC_HIP(hipMemcpyAsync(stage + off, src, ulen, hipMemcpyDeviceToHost, s0));
							jobs.push_back({rows[u], stage + off, ulen});
						}
					}
					if (sink->y && sink->y[u]) WC_HIP(hipMemcpyAsync(sink->y[u], d_y + yo[u], sizeof(double) * (size_t)y_len[u], hipMemcpyDeviceToHost, s0));
				}
				WC_HIP(hipStreamSynchronize(s0));
				parallel_copy(jobs);
			}
		}
		return WC_OK;
	};
	if (p->mode == 1 && n_utt >= 2) {
		// Two chains over the two halves of the batch, scheduled so that ALU-bound kernels never compete with each
		// other (measured: Harvest refinement and D4C merely time-share a CU) while every latency-bound stretch
		// runs in the shadow of an ALU-bound one:
		//   Harvest_A heavy | tail_A next to Harvest_B heavy | tail_B next to CheapTrick/D4C_A | pulses_A | D4C_B | pulses_B
		// (tail = unreliable-candidate test, contour logic, smoothing; the Synthesis time bases hide the same way)
		p->grp[0].main = s0;  // chain A runs on the calling thread's stream (wc_set_stream) -- resolved per call, not at creation
		// A device-resident batch runs as two halves.  A run whose rows leave for the host may ask for up to six groups of growing
		// size (HostSink::ng, ub): the same chain of chains -- group g's Harvest front behind group g - 1's refinement -- with every
		// group's CheapTrick / D4C on a high-priority stream as soon as its own contour is there.
		int NG = (sink && sink->ng >= 2 && sink->ng <= kMaxGroups && sink->ub[sink->ng] == n_utt) ? sink->ng : 2;
		int ub[kMaxGroups + 1];
		ub[0] = 0; ub[1] = n_utt / 2;
		for (int g = 2; g <= kMaxGroups; ++g) ub[g] = n_utt;
		if (NG > 2 || (sink && sink->ng == 2)) for (int g = 0; g <= NG; ++g) ub[g] = sink->ub[g];
		if (!sink) {  // WC_PIPELINE_GROUPS: a device-resident batch in that many equal groups (experiment)
			const int want = p->knobs->groups;
			if (want > 2 && want <= kMaxGroups && n_utt >= want) {
				NG = want;
				for (int g = 0; g <= NG; ++g) ub[g] = (int)((long long)n_utt * g / NG);
			}
		}
		if ((rc = pipeline_ensure_groups(p, NG))) return rc;
		const bool eager = sink && sink->eager;
		// streams: two groups as ever (own main and aux streams; the first group's aux is the high-priority one when its rows are
		// waited for).  More groups take turns on the two main streams -- group g + 2's Harvest sits behind group g's pulses there,
		// by which time group g + 1's refinement, its own start signal, is about through -- and share the high-priority stream for
		// CheapTrick / D4C, which run in group order anyway.
		hipStream_t aux[kMaxGroups], mainS[kMaxGroups];
		for (int g = 0; g < NG; ++g) {
			mainS[g] = p->grp[g & 1].main;
			if (NG == 2) aux[g] = (eager && g == 0 && p->grp[0].aux_hi) ? p->grp[0].aux_hi : p->grp[g].aux;
			else aux[g] = p->grp[0].aux_hi ? p->grp[0].aux_hi : p->grp[0].aux;
		}
		// Round 5: in a run of more than two groups the Synthesis of a group (time base, pulses) has a stream of its own -- the two
		// plain aux streams, which such a run does not use otherwise, in turn.  On the group's main stream it stood between the
		// Harvest of group g and that of group g + 2, which then could not start before group g's D4C was through and its pulses
		// summed: the contours of the later groups -- what the rows on the wire wait for -- came late.  WC_PIPELINE_SYN_STREAMS=0: as before.
		const bool syn_own = NG > 2 && p->grp[0].aux_hi && p->knobs->syn_streams != 0;
		hipStream_t synS[kMaxGroups];
		for (int g = 0; g < NG; ++g) synS[g] = syn_own ? ((g & 1) ? p->s2 : p->s1) : mainS[g];
		// WC_PIPELINE_CHAIN_MIN=n: a group's Harvest is not held behind that of a predecessor with fewer than n utterances (a group
		// of two or three fills a few per cent of the chip: holding the next one back behind it is latency for nothing)
		const int chain_min = p->knobs->chain_min;
		bool full[kMaxGroups][2] = {};
		double total_s = 0.0;
		for (int u = 0; u < n_utt; ++u) total_s += (double)x_length[u] / p->fs;
		for (int attempt = 0; attempt < 4; ++attempt) {
			const int bins_ = p->fft_size / 2 + 1;
			struct Slice { int u0, nu; long long xo, fo, yo; } sl[kMaxGroups];
			long long fo_end[kMaxGroups];
			{
				long long xo = 0, fo = 0, yo = 0;
				for (int g = 0; g < NG; ++g) {
					sl[g].u0 = ub[g];
					sl[g].nu = ub[g + 1] - ub[g];
					sl[g].xo = xo; sl[g].fo = fo; sl[g].yo = yo;
					for (int u = sl[g].u0; u < sl[g].u0 + sl[g].nu; ++u) { xo += x_length[u]; fo += f_len[u]; yo += y_len[u]; }
					fo_end[g] = fo;
				}
			}
			// 0. chain B's own stream starts behind whatever already sits on the caller's stream (an upload of the samples, the
			//    caller's kernels): its decimation reads d_x right away.  (The aux streams follow their main streams through e0.)
			// WC_PIPELINE_TIMING: where every group stands on the device's own clock (events with time stamps, development aid)
			static const bool gpu_marks = getenv("WC_PIPELINE_TIMING") != nullptr;
			hipEvent_t &tm0 = p->tm0;
			hipEvent_t (&tm)[kMaxGroups][4] = p->tm;
			if (gpu_marks && !tm0) {
				WC_HIP(hipEventCreate(&tm0));
				for (int g = 0; g < kMaxGroups; ++g) for (int k = 0; k < 4; ++k) WC_HIP(hipEventCreate(&tm[g][k]));
			}
			if (gpu_marks) WC_HIP(hipEventRecord(tm0, s0));
			WC_HIP(hipEventRecord(p->e1, s0));
			WC_HIP(hipStreamWaitEvent(mainS[1], p->e1, 0));
			// (a group's upload is waited for by its own Harvest, below: on a shared main stream the wait must not stand in front
			// of the earlier group's kernels)
			// 1. both Harvest chains; B's front starts when A's refinement kernel is done, A's tail runs beside it.
			// (WC_PIPELINE_TAIL_AFTER_BP=1, measured and rejected: A's tail held back until B's band-pass is through -- that kernel is one
			// round of long-lived wavefronts, 3040 on 3072 places at three per SIMD, and the places A's tail takes push some of them into
			// a second round, 2.6 instead of 1.8 ms.  But the tail then shares the chip with B's issue-bound raw candidates and refinement,
			// ends later and holds up A's CheapTrick: 29.2 against 28.6 ms per batch.  Enqueued front A, chain B, tail A in that case: an
			// event must have been recorded by the time a stream is told to wait for it.)
			const bool tail_late_env = p->knobs->tail_after_bp == 1;
			const bool tail_late = tail_late_env && NG == 2;
			// (WC_PIPELINE_CHAIN=0, measured and rejected: the Harvests of small neighbouring groups not held apart -- their full-grid
			// kernels do not fill the chip, a 6-utterance band-pass is one round of 650 wavefronts on 3072 places -- 52-60 ms
			// against 50: every group's contour arrives later)
			const bool chain_env = p->knobs->chain == 0;
			// Round 5: a device-resident batch of less than 500 s of signal runs its two groups' full-grid kernels SIDE BY SIDE -- neither
			// group fills the chip (the sliding band-pass of 32 x 10 s is one round of 3040 wavefronts on 3072 places: two halves of a
			// smaller batch fit that round together), and holding them apart is latency for nothing: 16 x 10 s 10.2 -> 9.0 ms,
			// 32 x 10 s 15.2 -> 14.1, 48 x 10 s 22.9 -> 21.9; 52 / 56 / 64 x 10 s are faster held apart (23.2 / 24.4 / 27.3 against
			// 23.7 / 25.0 / 28.0 ms; profiles/r05_d_small_batches_side_by_side.txt).  WC_PIPELINE_UNCHAIN_BELOW=seconds (0: never).
			const double unchain_below = p->knobs->unchain_below;
			const bool side_by_side = NG == 2 && !sink && total_s < unchain_below;
			// (experiment, WC_PIPELINE_SIDE=h / c: only the Harvests, or only the first group's CheapTrick / D4C and the second's Harvest, side by side)
			const bool side_h = side_by_side || (NG == 2 && !sink && p->knobs->side == 'h');
			const bool side_c = side_by_side || (NG == 2 && !sink && p->knobs->side == 'c');
			const bool chain_harvest = (NG == 2 && !side_h) || (NG > 2 && !chain_env);
			auto enqueue_harvest = [&](int g) -> int {
				PipeGroup &G = p->grp[g];
				dev->time_tag = g;
				if (g >= 1 && sink && sink->x_ev[g]) WC_HIP(hipStreamWaitEvent(mainS[g], sink->x_ev[g], 0));
				wc_harvest *hvg = G.hv;
				return hv_enqueue(hvg, mainS[g], sl[g].nu, d_x + sl[g].xo, x_length + sl[g].u0, d_tpos + sl[g].fo, d_f0 + sl[g].fo,
								  full[g][0], G.e_mid, (g >= 1 && chain_harvest && sl[g - 1].nu >= chain_min) ? p->grp[g - 1].e_mid : nullptr, (g == 0 && tail_late) ? 1 : 3,
								  (g == 1 && tail_late) ? p->e_bp : nullptr, nullptr);
			};
			// (two groups: both Harvest chains first -- the first group's CheapTrick / D4C may wait for an event of the second's.
			// More groups share streams: a group's whole chain is enqueued before the next group's Harvest lands on its stream.)
			// Round 5, the default for a device-resident batch that fills the chip (500 s of signal and more): TWO LANES instead of four
			// streams.  Every full-grid kernel of the step on one stream (the caller's), in the order front_A, front_B, CheapTrick /
			// D4C_A, pulses_A, CheapTrick / D4C_B, pulses_B -- none of them shares the chip with another one --; every latency-bound
			// stretch (a group's candidate test, contour logic and smoothing, CheapTrick's frame counts, the Synthesis time base) on the
			// other, behind the front it needs and in front of the kernels that need it.  The four-stream order below (two chains
			// held apart by events) reaches the same 27.1-27.3 ms per 64 x 10 s on boxes whose queue scheduler happens to deal its
			// streams well and 28.3-28.7 ms on others (a third of the round's boxes); this one measured 27.2-27.3 on both kinds
			// (profiles/r05_e_two_lanes.txt).  Same bits.  WC_PIPELINE_SCHEDULE=chains / option "schedule": the four-stream order
			// (WC_PIPELINE_TIMING's device marks are recorded by that order only).
			const bool lanes = NG == 2 && !sink && !side_by_side && !p->knobs->schedule_chains;
			if (lanes) {
				// (measured and dropped: the small kernels on the high-priority stream, 27.7 against 27.3 ms; the second group's front on a
				// stream of its own so that its decimation runs underneath the first front, 27.8-28.0)
				const hipStream_t F = mainS[0], S = p->grp[0].aux;
				WC_HIP(hipStreamWaitEvent(S, p->e1, 0));  // (behind whatever precedes this call on the caller's stream)
				wc_harvest *hvg[2];
				for (int g = 0; g < 2; ++g) {
					hvg[g] = p->grp[g].hv;
					dev->time_tag = g;
					// Round 6, measured and NOT kept (option "pre_lane" = 1 / 2): what the second group's front needs nothing but the samples
					// for -- decimation, DC, level marks, seam values: 0.45 ms of kernels that leave most of the chip idle between the two
					// groups' full-grid kernels -- on the latency lane (or a stream of its own) behind the first group's band-pass.  Under
					// rocprofv3's kernel trace the step is 0.24 ms shorter that way (the second band-pass starts 0.29 ms earlier); without
					// the tracer, timed alternately in one process, it is 0.6 ms LONGER (27.00 / 27.05 against 26.39 ms,
					// profiles/r06_b_pre_lane_ab.txt): the decimation's workgroups hold 67 KB of LDS each and take the places of two of the
					// refinement's four workgroups on the CUs they land on.
					const bool pre_on_s = g == 1 && p->knobs->pre_lane != 0;
					if (pre_on_s) {
						// (behind the first group's band-pass: that kernel is one round of long-lived wavefronts on nearly every place of
						// the chip, and whatever takes places beside it pushes some of them into a second round -- 26.7 against 26.3 ms)
						const hipStream_t P = p->knobs->pre_lane == 2 ? p->grp[1].aux : S;  // (2: a stream of its own)
						if (P != S) WC_HIP(hipStreamWaitEvent(P, p->e1, 0));
						WC_HIP(hipStreamWaitEvent(P, p->e2, 0));
						if ((rc = hv_enqueue(hvg[g], P, sl[g].nu, d_x + sl[g].xo, x_length + sl[g].u0, d_tpos + sl[g].fo, d_f0 + sl[g].fo, full[g][0],
											 nullptr, nullptr, 1 | 4, nullptr, nullptr)))
							return rc;
						WC_HIP(hipEventRecord(p->e_bp, P));
					}
					if ((rc = hv_enqueue(hvg[g], F, sl[g].nu, d_x + sl[g].xo, x_length + sl[g].u0, d_tpos + sl[g].fo, d_f0 + sl[g].fo, full[g][0],
										 p->grp[g].e_mid, pre_on_s ? p->e_bp : nullptr, pre_on_s ? (1 | 8) : 1, (g == 0 && p->knobs->pre_lane != 0) ? p->e2 : nullptr, nullptr)))
						return rc;
				}
				for (int g = 0; g < 2; ++g) {
					PipeGroup &G = p->grp[g];
					dev->time_tag = g;
					const int u0 = sl[g].u0, nu = sl[g].nu;
					const double *gx = d_x + sl[g].xo;
					double *gt = d_tpos + sl[g].fo, *gf = d_f0 + sl[g].fo, *gsp = d_sp + sl[g].fo * bins_, *gap = d_ap + sl[g].fo * bins_;
					double *gy = d_y + sl[g].yo;
					const uint64_t *grp_rng = rng_start ? rng_start + u0 : nullptr;
					long long total = 0;
					uint64_t a0 = 0, a1 = 0;
					if ((rc = hv_enqueue(hvg[g], S, nu, gx, x_length + u0, gt, gf, full[g][0], nullptr, nullptr, 2, nullptr, G.e_mid))) return rc;
					if ((rc = ct_prepare(G.ct, S, nu, x_length + u0, gf, f_len.data() + u0, grp_rng, &total, &a0, &a1))) return rc;
					WC_HIP(hipEventRecord(G.e0, S));
					if ((rc = syn_prepare(G.sy, S, nu, gf, f_len.data() + u0, y_len.data() + u0, gy, nullptr, full[g][1]))) return rc;
					WC_HIP(hipEventRecord(G.e_ct, S));
					WC_HIP(hipStreamWaitEvent(F, G.e0, 0));
					hipEvent_t ct_rows = nullptr;
					if ((rc = ct_frames(G.ct, F, nu, gx, gt, gf, gsp, total, &ct_rows))) return rc;
					if ((rc = d4c_enqueue(G.d4, F, nu, gx, x_length + u0, gt, gf, f_len.data() + u0, p->fft_size, gap, nullptr, ct_end_positions(G.ct))))
						return rc;
					// (measured and dropped: the pulses on a stream of their own, the first group's beside the second group's CheapTrick /
					// D4C as in the four-stream order: 27.2-27.3 ms either way, 42.0 against 40.5 ms at 96 x 10 s)
					if (ct_rows) WC_HIP(hipStreamWaitEvent(F, ct_rows, 0));
					WC_HIP(hipStreamWaitEvent(F, G.e_ct, 0));
					if ((rc = syn_pulses(G.sy, F, gf, gsp, gap, gy, d4c_end_positions(G.d4)))) return rc;
				}
				dev->time_tag = -1;
				bool again = false;
				std::vector<int> tied;
				for (int g = 0; g < 2; ++g) {
					bool o1 = false, o2 = false, tie = false;
					std::vector<int> tg;
					if ((rc = syn_finish(p->grp[g].sy, F, rng_pos ? rng_pos + ub[g] : nullptr, &o2))) return rc;
					if ((rc = hv_overflowed(hvg[g], F, &o1, &tie, &tg))) return rc;
					full[g][0] = full[g][0] || o1;
					full[g][1] = full[g][1] || o2;
					for (int u : tg) tied.push_back(ub[g] + u);
					again = again || o1 || o2;
				}
				if (!again) return fix_up(tied);
				continue;
			}
			if (NG == 2) for (int g = 0; g < NG; ++g) if ((rc = enqueue_harvest(g))) return rc;
			if (tail_late) {
				PipeGroup &G = p->grp[0];
				dev->time_tag = 0;
				if ((rc = hv_enqueue(G.hv, mainS[0], sl[0].nu, d_x + sl[0].xo, x_length + sl[0].u0, d_tpos + sl[0].fo, d_f0 + sl[0].fo,
									 full[0][0], nullptr, nullptr, 2, nullptr, p->e_bp)))
					return rc;
			}
			// 2. the rest of each chain; A's CheapTrick/D4C wait for B's refinement as well
			for (int g = 0; g < NG; ++g) {
				if (NG > 2 && (rc = enqueue_harvest(g))) return rc;
				PipeGroup &G = p->grp[g];
				const hipStream_t G_aux = aux[g];
				dev->time_tag = g;
				const int u0 = sl[g].u0, nu = sl[g].nu;
				const double *gx = d_x + sl[g].xo;
				double *gt = d_tpos + sl[g].fo, *gf = d_f0 + sl[g].fo, *gsp = d_sp + sl[g].fo * bins_, *gap = d_ap + sl[g].fo * bins_;
				double *gy = d_y + sl[g].yo;
				const uint64_t *grp_rng = rng_start ? rng_start + u0 : nullptr;
				long long total = 0;
				uint64_t a0 = 0, a1 = 0;
				if ((rc = ct_prepare(G.ct, mainS[g], nu, x_length + u0, gf, f_len.data() + u0, grp_rng, &total, &a0, &a1))) return rc;
				WC_HIP(hipEventRecord(G.e0, mainS[g]));
				if (gpu_marks) WC_HIP(hipEventRecord(tm[g][0], mainS[g]));
				WC_HIP(hipStreamWaitEvent(G_aux, G.e0, 0));
				// (a run whose rows leave for the host is bound by PCIe, not by the kernels: there the first half's rows are wanted
				// as early as they can be had, even if its CheapTrick / D4C then share the CUs with the second half's Harvest)
				if (NG == 2 && g == 0 && !eager && !side_c) WC_HIP(hipStreamWaitEvent(G_aux, p->grp[1].e_mid, 0));
				hipEvent_t ct_rows = nullptr;  // CheapTrick's pass over the frames its one-wavefront kernel leaves out, on a stream of its own
				if ((rc = ct_frames(G.ct, G_aux, nu, gx, gt, gf, gsp, total, &ct_rows))) return rc;
				WC_HIP(hipEventRecord(G.e_ct, G_aux));
				if (gpu_marks) WC_HIP(hipEventRecord(tm[g][1], G_aux));
				if ((rc = d4c_enqueue(G.d4, G_aux, nu, gx, x_length + u0, gt, gf, f_len.data() + u0, p->fft_size, gap, nullptr,
									  ct_end_positions(G.ct))))
					return rc;
				if (ct_rows) WC_HIP(hipStreamWaitEvent(G_aux, ct_rows, 0));  // (the pulses wait for e_aux)
				WC_HIP(hipEventRecord(G.e_aux, G_aux));
				if (gpu_marks) WC_HIP(hipEventRecord(tm[g][2], G_aux));
				if (sink && attempt == 0 && (sink->stage_sp || sink->stage_ap)) {
					const size_t off = sizeof(double) * (size_t)sl[g].fo * bins_, len = sizeof(double) * (size_t)(fo_end[g] - sl[g].fo) * bins_;
					// the spectrogram rows leave as soon as CheapTrick is through, the aperiodicity rows behind D4C: PCIe is the longest
					// stretch of a run with all outputs (2.1 GB at ~50 GB/s), so it starts as early as it can
					// (direct rows: each array's utterances are dealt to the two copy streams alternately, so that both engines work
					// on whatever is ready and finish together; staged: spectrogram on one stream, aperiodicity on the other)
					for (int which = 0; which < 2; ++which) {
						char *stage = which == 0 ? sink->stage_sp : sink->stage_ap;
						double *const *rows = which == 0 ? sink->sp : sink->ap;
						const double *src = which == 0 ? gsp : gap;
						if (!stage) continue;
						for (int lane = 0; lane < 2; ++lane) {
							hipStream_t sc = (lane == 1 && p->s_copy2) ? p->s_copy2 : p->s_copy;
							if (lane == 1 && !p->s_copy2) break;
							WC_HIP(hipStreamWaitEvent(sc, which == 0 ? G.e_ct : G.e_aux, 0));
							if (which == 0 && ct_rows) WC_HIP(hipStreamWaitEvent(sc, ct_rows, 0));
						}
						if (sink->direct) {
							long long fo2 = 0;
							for (int u = u0; u < u0 + nu; ++u) {
								const size_t ulen = sizeof(double) * (size_t)f_len[u] * bins_;
								hipStream_t sc = (p->s_copy2 && ((u - u0) & 1)) ? p->s_copy2 : p->s_copy;
								if (rows[u]) WC_HIP(hipMemcpyAsync(rows[u], src + fo2 * bins_, ulen, hipMemcpyDeviceToHost, sc));
								fo2 += f_len[u];
							}
						} else {
							hipStream_t sc = (which == 1 && p->s_copy2) ? p->s_copy2 : p->s_copy;
							WC_HIP(hipMemcpyAsync(stage + off, src, len, hipMemcpyDeviceToHost, sc));
						}
					}
					if (p->s_copy2) WC_HIP(hipEventRecord(p->e_copy2[g], p->s_copy2));
					WC_HIP(hipEventRecord(p->e_copy[g], p->s_copy));
				}
				if (syn_own) WC_HIP(hipStreamWaitEvent(synS[g], G.e0, 0));  // (the contour; e0 stands behind ct_prepare on the main stream)
				if ((rc = syn_prepare(G.sy, synS[g], nu, gf, f_len.data() + u0, y_len.data() + u0, gy, nullptr, full[g][1]))) return rc;
				WC_HIP(hipStreamWaitEvent(synS[g], G.e_aux, 0));
				if ((rc = syn_pulses(G.sy, synS[g], gf, gsp, gap, gy, d4c_end_positions(G.d4)))) return rc;
				if (gpu_marks) WC_HIP(hipEventRecord(tm[g][3], synS[g]));
				if (sink && sink->y) {
					// (every attempt: a re-run after an overflow rewrites the waveform, and its copies land behind the first ones)
					WC_HIP(hipEventRecord(p->e_y[g], synS[g]));
					WC_HIP(hipStreamWaitEvent(p->s_copy, p->e_y[g], 0));
					if (p->s_copy2) WC_HIP(hipStreamWaitEvent(p->s_copy2, p->e_y[g], 0));
					long long yo2 = 0;
					for (int u = u0; u < u0 + nu; ++u) {
						hipStream_t sc = (p->s_copy2 && ((u - u0) & 1)) ? p->s_copy2 : p->s_copy;
						if (sink->y[u]) WC_HIP(hipMemcpyAsync(sink->y[u], gy + yo2, sizeof(double) * (size_t)y_len[u], hipMemcpyDeviceToHost, sc));
						yo2 += y_len[u];
					}
					if (p->s_copy2) {  // (one event for both streams: the first waits for the second)
						WC_HIP(hipEventRecord(p->e_ycopy[g], p->s_copy2));
						WC_HIP(hipStreamWaitEvent(p->s_copy, p->e_ycopy[g], 0));
					}
					WC_HIP(hipEventRecord(p->e_ycopy[g], p->s_copy));
				}
			}
			dev->time_tag = -1;
			pmark("both halves enqueued");
			if (sink && attempt == 0 && (sink->stage_sp || sink->stage_ap)) {
				// the rows of each half batch go to the caller's buffers as soon as their copy has landed: half A's while B computes
				for (int g = 0; g < NG; ++g) {
					WC_HIP(hipEventSynchronize(p->e_copy[g]));
					if (p->s_copy2) WC_HIP(hipEventSynchronize(p->e_copy2[g]));
					pmark(g == 0 ? "rows of group 0 landed" : g + 1 < NG ? "rows of a middle group landed" : "rows of the last group landed");
					if (sink->direct) { sink->overlapped[g] = true; continue; }
					std::vector<CopyJob> jobs;
					long long fo = sl[g].fo;
					for (int u = sl[g].u0; u < sl[g].u0 + sl[g].nu; ++u) {
						const size_t off = sizeof(double) * (size_t)fo * bins_, len = sizeof(double) * (size_t)f_len[u] * bins_;
						if (sink->stage_sp && sink->sp[u]) jobs.push_back({sink->sp[u], sink->stage_sp + off, len});
						if (sink->stage_ap && sink->ap[u]) jobs.push_back({sink->ap[u], sink->stage_ap + off, len});
						fo += f_len[u];
					}
					parallel_copy(jobs);
					sink->overlapped[g] = true;
				}
			}
			bool again = false;
			std::vector<int> tied;
			for (int g = 0; g < NG; ++g) {
				PipeGroup &G = p->grp[g];
				const int u0 = ub[g];
				bool o1 = false, o2 = false;
				if ((rc = syn_finish(G.sy, synS[g], rng_pos ? rng_pos + u0 : nullptr, &o2))) return rc;
				pmark(g == 0 ? "group 0 finished" : g + 1 < NG ? "a middle group finished" : "the last group finished");
				bool tie = false;
				std::vector<int> tg;
				if ((rc = hv_overflowed(G.hv, mainS[g], &o1, &tie, &tg))) return rc;
				full[g][0] = full[g][0] || o1;
				full[g][1] = full[g][1] || o2;
				for (int u : tg) tied.push_back(u0 + u);
				again = again || o1 || o2;
			}
			if (again && sink) for (int g = 0; g < kMaxGroups; ++g) sink->overlapped[g] = false;  // the re-run rewrites the rows
			if (!again && sink && sink->y) {
				for (int g = 0; g < NG; ++g) WC_HIP(hipEventSynchronize(p->e_ycopy[g]));
				sink->y_done = true;
				pmark("waveforms landed");
			}
			if (gpu_marks) {
				(void)hipDeviceSynchronize();
				for (int g = 0; g < NG; ++g) {
					float t[4] = {0, 0, 0, 0};
					for (int k = 0; k < 4; ++k) (void)hipEventElapsedTime(&t[k], tm0, tm[g][k]);
					std::fprintf(stderr, "  [pipeline] group %d (%d utterances) on the device: contour %.2f  CheapTrick %.2f  D4C %.2f  pulses %.2f ms\n",
								 g, ub[g + 1] - ub[g], t[0], t[1], t[2], t[3]);
				}
			}
			if (!again) return fix_up(tied);
		}
		return fail(WC_ERR_DEVICE, "pipeline: buffer overflow");
	}
	bool hv_full = false, syn_full = false;
	const int ns = p->n_split < n_utt ? p->n_split : n_utt;
	for (int attempt = 0; attempt < 4; ++attempt) {
		{
			// group k = utterances [u0, u1); the packed layout makes every group a contiguous slice
			long long xo = 0, fo = 0;
			if (ns > 1) WC_HIP(hipEventRecord(p->e0, s0));  // later groups start after whatever precedes on s0
			for (int k = 0; k < ns; ++k) {
				const int u0 = (int)((long long)n_utt * k / ns), u1 = (int)((long long)n_utt * (k + 1) / ns);
				hipStream_t sk = k == 0 ? s0 : p->hs[k];
				if (k > 0) WC_HIP(hipStreamWaitEvent(sk, p->e0, 0));
				wc_harvest *hvk = p->hv[k];
				if ((rc = hv_enqueue(hvk, sk, u1 - u0, d_x + xo, x_length + u0, d_tpos + fo, d_f0 + fo, hv_full, nullptr, nullptr))) return rc;
				if (k > 0) WC_HIP(hipEventRecord(p->he[k], sk));
				for (int u = u0; u < u1; ++u) { xo += x_length[u]; fo += f_len[u]; }
			}
			for (int k = 1; k < ns; ++k) WC_HIP(hipStreamWaitEvent(s0, p->he[k], 0));
		}
		long long total = 0;
		uint64_t a0 = 0, a1 = 0;
		if ((rc = ct_prepare(p->ct, s0, n_utt, x_length, d_f0, f_len.data(), rng_start, &total, &a0, &a1))) return rc;
		WC_HIP(hipEventRecord(p->e0, s0));
		WC_HIP(hipStreamWaitEvent(ns > 1 ? p->hs[1] : p->s1, p->e0, 0));
		WC_HIP(hipStreamWaitEvent(p->s2, p->e0, 0));
		// CheapTrick's frames go to the second Harvest stream when there is one: HIP multiplexes streams onto a few
		// hardware queues, and two streams that land on the same queue would serialise CheapTrick and D4C
		hipStream_t sct = ns > 1 ? p->hs[1] : p->s1;
		if ((rc = ct_frames(p->ct, sct, n_utt, d_x, d_tpos, d_f0, d_sp, total, nullptr))) return rc;
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
		std::vector<int> tied;
		for (int k = 0; k < ns; ++k) {
			bool ok = false, tie = false;
			std::vector<int> tg;
			if ((rc = hv_overflowed(p->hv[k], s0, &ok, &tie, &tg))) return rc;  // s0 is already idle; the flag copies are tiny
			o1 = o1 || ok;
			for (int u : tg) tied.push_back((int)((long long)n_utt * k / ns) + u);  // (like the groups of the chained schedule above)
		}
		if (!o1 && !o2) return fix_up(tied);
		hv_full = hv_full || o1;
		syn_full = syn_full || o2;
	}
	return fail(WC_ERR_DEVICE, "pipeline: buffer overflow");
}

int wc_pipeline_run_device(wc_pipeline *p, int n_utt, const double *d_x, const int *x_length, double *d_tpos, double *d_f0,
						   double *d_sp, double *d_ap, double *d_y, uint64_t *rng_pos) {
	return pipeline_run(p, n_utt, d_x, x_length, d_tpos, d_f0, d_sp, d_ap, d_y, rng_pos, nullptr);
}

// Host batch front-end: ragged utterances as 16-bit PCM (as stored in a WAV file) or as doubles,
// gathered into pinned memory, one H2D copy, expanded on the device, fused pipeline, outputs packed into pinned memory with
// one D2H copy per requested array, scattered to the caller's per-utterance buffers.  Any output table may be NULL.
int wc_pipeline_run_batch_host(wc_pipeline *p, int n_utt, const void *const *x, int x_is_pcm16, const int *x_length, double *const *tpos,
							   double *const *f0, double *const *sp, double *const *ap, void *const *y, int y_is_pcm16,
							   uint64_t *rng_pos) {
	if (!p || n_utt <= 0 || !x || !x_length) return fail(WC_ERR_INVALID, "pipeline batch: null argument");
	WC_HIP(hipSetDevice(p->dev->id));
	DeviceLock lock(p->dev);
	hipStream_t s = p->dev->active();
	const int bins = p->fft_size / 2 + 1;
	// WC_PIPELINE_TIMING=1: wall-clock marks of the phases on stderr (development aid)
	g_mark0 = std::chrono::steady_clock::now();
	auto mark = [&](const char *what) { pmark(what); };
	std::vector<int> f_len(n_utt), y_len(n_utt);
	long long nx = 0, nf = 0, ny = 0;
	for (int u = 0; u < n_utt; ++u) {
		if (x_length[u] <= 0 || !x[u]) return fail(WC_ERR_INVALID, "pipeline batch: empty utterance");
		f_len[u] = wc_get_samples(p->fs, x_length[u], p->frame_period);
		y_len[u] = wc_synthesis_out_length(f_len[u], p->frame_period, p->fs);
		nx += x_length[u]; nf += f_len[u]; ny += y_len[u];
	}
	int rc;
	if (x_is_pcm16 < 0 || x_is_pcm16 > 2) return fail(WC_ERR_INVALID, "pipeline batch: input format must be 0 (float64), 1 (int16 PCM) or 2 (float32)");
	const size_t in_elem = x_is_pcm16 == 1 ? sizeof(int16_t) : x_is_pcm16 == 2 ? sizeof(float) : sizeof(double);
	if ((rc = p->st_in.reserve(in_elem * nx))) return rc;
	if ((rc = p->b_x.reserve(sizeof(double) * nx))) return rc;
	if (x_is_pcm16 && (rc = p->b_pcm.reserve(in_elem * nx))) return rc;
	if ((rc = p->b_t.reserve(sizeof(double) * nf))) return rc;
	if ((rc = p->b_f.reserve(sizeof(double) * nf))) return rc;
	if ((rc = p->b_sp.reserve(sizeof(double) * nf * bins))) return rc;
	if ((rc = p->b_ap.reserve(sizeof(double) * nf * bins))) return rc;
	if ((rc = p->b_y.reserve(sizeof(double) * ny))) return rc;
	// The samples go up group by group: the first group's Harvest starts behind its own upload while the others'
	// are still on the wire (copy stream).  Page-locked utterances are read by the copy engine where they lie; others
	// are gathered into pinned staging by a few threads (245 MB of doubles take 40 ms on one).
	HostSink sink;
	{
		// A run whose spectrogram / aperiodicity rows leave for the host is bound by PCIe (2.1 GB per 64 x 10 s at 48 kHz: 37 ms at
		// 57 GB/s against 28 ms of kernels): what counts is how early the FIRST rows are ready and that the copy engines never wait
		// afterwards.  A first group of a third of the batch has its rows ready earlier, and they take about as long to copy as the
		// rest of the batch needs to get its own ready (WC_PIPELINE_HOST_SPLIT: per cent of the utterances in the first group;
		// measured on 64 x 10 s at 48 kHz, all five outputs: 50 % 59.0 ms, 35 % 55.9 ms, 25 % 57.3 ms, 12 % 60.8 ms).
		// Round 4: more groups of growing size do better still.  The kernels get an utterance ready in 0.44 ms, the link takes 0.64 ms
		// for its rows: once the first group's rows are on the wire the copy engines never wait again as long as group g is no larger
		// than the first group plus 0.45 of everything before it.  Measured (64 x 10 s at 48 kHz, all five outputs, this box): two groups
		// 35 / 65 %: 55.2 ms; three 20 / 30 / 50: 51.3; four 12 / 20 / 30 / 38: 49.9; five 10 / 13 / 18 / 25 / 34: 49.6; six 6 / 8 / 11 / 15 / 21 / 39:
		// 51.0 (41 ms for the bytes; the first rows cannot leave before ~7 ms whatever the first group's size: Harvest's tail and the
		// first CheapTrick are latency, not work).  WC_PIPELINE_HOST_SPLITS: per cent of the utterances in every group but the last.
		int ng = (p->mode == 1 && n_utt >= 2) ? 2 : 1;
		int ub[kMaxGroups + 1];
		ub[0] = 0; ub[1] = n_utt / 2;
		for (int g = 2; g <= kMaxGroups; ++g) ub[g] = n_utt;
		if (p->mode == 1 && n_utt >= 4 && (sp || ap)) {
			std::vector<int> pct;
			if (p->knobs->host_splits_set) {
				pct = p->knobs->host_splits;
			} else if (n_utt >= 32) {
				pct = {5, 8, 12, 18, 25};  // round 5 (with Synthesis on streams of its own): 3 / 5 / 8 / 11 / 16 / 21 of 64 utterances
			} else if (n_utt >= 20) {
				pct = {10, 13, 18, 25};
			} else if (n_utt >= 8) {
				pct = {20, 30};
			} else {
				pct = {35};
			}
			if ((int)pct.size() > n_utt - 1) pct.resize(n_utt - 1);  // (every group gets at least one utterance)
			ng = (int)pct.size() + 1;
			int acc = 0;
			for (int g = 0; g + 1 < ng; ++g) {
				acc += pct[g];
				// (every group gets at least one utterance)
				ub[g + 1] = std::min(n_utt - (ng - 1 - g), std::max(ub[g] + 1, (int)((long long)n_utt * acc / 100)));
			}
			ub[ng] = n_utt;
			sink.ng = ng;
			for (int g = 0; g <= ng; ++g) sink.ub[g] = ub[g];
		}
		const bool halves = ng >= 2 && x_is_pcm16 == 0;
		bool pinned_in = halves && p->knobs->direct != 0;
		for (int u = 0; u < n_utt && pinned_in; ++u) pinned_in = is_pinned(x[u]);
		char *dst = static_cast<char *>(p->st_in.p);
		long long xo = 0;
		for (int part = 0; part < (halves ? ng : 1); ++part) {
			const int u0 = halves ? ub[part] : 0, u1 = halves ? ub[part + 1] : n_utt;
			hipStream_t sc = part == 0 ? s : p->s_copy;
			long long n_part = 0;
			for (int u = u0; u < u1; ++u) n_part += x_length[u];
			if (pinned_in) {
				long long o = xo;
				for (int u = u0; u < u1; ++u) {
					WC_HIP(hipMemcpyAsync(p->b_x.as<double>() + o, x[u], sizeof(double) * (size_t)x_length[u], hipMemcpyHostToDevice, sc));
					o += x_length[u];
				}
			} else {
				std::vector<CopyJob> jobs;
				char *d0 = dst;
				for (int u = u0; u < u1; ++u) {
					jobs.push_back({dst, x[u], in_elem * x_length[u]});
					dst += in_elem * x_length[u];
				}
				parallel_copy(jobs);
				if (part == 0) mark("inputs gathered");
				if (x_is_pcm16 == 1) {
					WC_HIP(hipMemcpyAsync(p->b_pcm.p, p->st_in.p, sizeof(int16_t) * nx, hipMemcpyHostToDevice, s));
					if ((rc = wc_pcm16_to_double_device(p->b_pcm.as<int16_t>(), nx, p->b_x.as<double>()))) return rc;
				} else if (x_is_pcm16 == 2) {
					WC_HIP(hipMemcpyAsync(p->b_pcm.p, p->st_in.p, sizeof(float) * nx, hipMemcpyHostToDevice, s));
					if ((rc = wc_float_to_double_device(p->b_pcm.as<float>(), nx, p->b_x.as<double>()))) return rc;
				} else {
					WC_HIP(hipMemcpyAsync(p->b_x.as<double>() + xo, d0, sizeof(double) * (size_t)n_part, hipMemcpyHostToDevice, sc));
				}
			}
			if (part >= 1) {
				WC_HIP(hipEventRecord(p->e_x[part], sc));
				sink.x_ev[part] = p->e_x[part];
			}
			xo += n_part;
		}
	}
	mark("inputs on their way");
	// outputs: one packed region of pinned memory, [tpos | f0 | sp | ap | y]
	const size_t y_elem = y_is_pcm16 ? sizeof(int16_t) : sizeof(double);
	size_t off_t = 0, off_f = off_t + (tpos ? sizeof(double) * nf : 0), off_sp = off_f + (f0 ? sizeof(double) * nf : 0);
	size_t off_ap = off_sp + (sp ? sizeof(double) * nf * bins : 0), off_y = off_ap + (ap ? sizeof(double) * nf * bins : 0);
	const size_t total = off_y + (y ? y_elem * ny : 0);
	if (total > 0 && (rc = p->st_out.reserve(total))) return rc;
	char *out = static_cast<char *>(p->st_out.p);
	// The big arrays (spectrogram, aperiodicity) leave per half batch while the rest of the batch still computes, and are
	// handed to the caller's rows by several threads (pipeline_run, HostSink).
	sink.stage_sp = sp ? out + off_sp : nullptr;
	sink.stage_ap = ap ? out + off_ap : nullptr;
	sink.sp = sp; sink.ap = ap; sink.f_len = f_len.data(); sink.bins = bins;
	// a caller who hands over page-locked rows (hipHostMalloc / hipHostRegister, a pinned torch tensor) gets them written by the
	// copy engine directly: no staging copy, no host-side scatter of the 2 GB
	{
		sink.direct = (sp || ap) && p->knobs->direct != 0;  // ("direct" = 0: always through the staging buffer, A/B)
	}
	for (int u = 0; u < n_utt && sink.direct; ++u)
		sink.direct = (!sp || !sp[u] || is_pinned(sp[u])) && (!ap || !ap[u] || is_pinned(ap[u]));
	{
		sink.eager = (sp || ap) && p->knobs->eager != 0;  // ("eager" = 0: the device-resident schedule, A/B)
	}
	if (y && !y_is_pcm16 && p->mode == 1 && n_utt >= 2 && p->knobs->direct != 0) {
		bool all = true;
		for (int u = 0; u < n_utt && all; ++u) all = !y[u] || is_pinned(y[u]);
		if (all) {
			sink.y = reinterpret_cast<double *const *>(y);
			sink.y_len = y_len.data();
		}
	}
	if ((rc = pipeline_run(p, n_utt, p->b_x.as<double>(), x_length, p->b_t.as<double>(), p->b_f.as<double>(),
						   p->b_sp.as<double>(), p->b_ap.as<double>(), p->b_y.as<double>(), rng_pos, &sink))) {
		// (a failed run may leave the second half's upload out of st_in, or row copies into the caller's page-locked buffers, in
		// flight on the copy streams: nothing of this call may still be moving when the caller gets its buffers back)
		(void)hipDeviceSynchronize();
		return rc;
	}
	mark("pipeline_run returned");
	if ((rc = p->st_in.mark(s))) return rc;  // (both uploads are long done: the run has been waited for)
	if (total == 0) return WC_OK;
	bool rows_done = true;
	for (int g = 0; g < (sink.ng >= 2 ? sink.ng : 2); ++g) rows_done = rows_done && sink.overlapped[g];
	if (tpos) WC_HIP(hipMemcpyAsync(out + off_t, p->b_t.p, sizeof(double) * nf, hipMemcpyDeviceToHost, s));
	if (f0) WC_HIP(hipMemcpyAsync(out + off_f, p->b_f.p, sizeof(double) * nf, hipMemcpyDeviceToHost, s));
	if (sp && !rows_done) WC_HIP(hipMemcpyAsync(out + off_sp, p->b_sp.p, sizeof(double) * nf * bins, hipMemcpyDeviceToHost, s));
	if (ap && !rows_done) WC_HIP(hipMemcpyAsync(out + off_ap, p->b_ap.p, sizeof(double) * nf * bins, hipMemcpyDeviceToHost, s));
	const bool y_staged = y && !sink.y_done;
	if (y_staged) {
		if (y_is_pcm16) {
			if ((rc = p->b_ypcm.reserve(sizeof(int16_t) * ny))) return rc;
			if ((rc = wc_double_to_pcm16_device(p->b_y.as<double>(), ny, p->b_ypcm.as<int16_t>()))) return rc;
			WC_HIP(hipMemcpyAsync(out + off_y, p->b_ypcm.p, sizeof(int16_t) * ny, hipMemcpyDeviceToHost, s));
		} else {
			WC_HIP(hipMemcpyAsync(out + off_y, p->b_y.p, sizeof(double) * ny, hipMemcpyDeviceToHost, s));
		}
	}
	WC_HIP(hipStreamSynchronize(s));
	mark("small outputs on the host");
	std::vector<CopyJob> jobs;
	long long fo = 0, yo = 0;
	for (int u = 0; u < n_utt; ++u) {
		if (tpos && tpos[u]) jobs.push_back({tpos[u], out + off_t + sizeof(double) * fo, sizeof(double) * f_len[u]});
		if (f0 && f0[u]) jobs.push_back({f0[u], out + off_f + sizeof(double) * fo, sizeof(double) * f_len[u]});
		if (!rows_done) {
			if (sp && sp[u]) jobs.push_back({sp[u], out + off_sp + sizeof(double) * fo * bins, sizeof(double) * f_len[u] * bins});
			if (ap && ap[u]) jobs.push_back({ap[u], out + off_ap + sizeof(double) * fo * bins, sizeof(double) * f_len[u] * bins});
		}
		if (y_staged && y[u]) jobs.push_back({y[u], out + off_y + y_elem * yo, y_elem * y_len[u]});
		fo += f_len[u];
		yo += y_len[u];
	}
	parallel_copy(jobs);
	mark("scattered");
	return WC_OK;
}

// The same front-end with the reference's feature codec as the epilogue (reference src/codec.cpp:211-325, world_class_codec.h):
// the spectral envelope leaves as number_of_dimensions mel-cepstral coefficients per frame, the aperiodicity as its
// GetNumberOfAperiodicities(fs) band values -- 65 instead of 2050 doubles per 48 kHz frame cross PCIe.  coded_sp[u]: f_len[u] x
// number_of_dimensions doubles, coded_ap[u]: f_len[u] x GetNumberOfAperiodicities(fs) doubles; f0 / tpos / y as above; any table
// may be NULL.
int wc_pipeline_run_batch_host_coded(wc_pipeline *p, int n_utt, const void *const *x, int x_is_pcm16, const int *x_length, double *const *tpos,
									 double *const *f0, double *const *coded_sp, int number_of_dimensions, double *const *coded_ap,
									 void *const *y, int y_is_pcm16, uint64_t *rng_pos) {
	if (!p || n_utt <= 0 || !x || !x_length) return fail(WC_ERR_INVALID, "pipeline batch: null argument");
	if (coded_sp && number_of_dimensions <= 0) return fail(WC_ERR_INVALID, "pipeline batch: number_of_dimensions must be positive");
	// everything but the big rows through the plain front-end (which keeps the batch resident in p->b_*)
	int rc = wc_pipeline_run_batch_host(p, n_utt, x, x_is_pcm16, x_length, tpos, f0, nullptr, nullptr, y, y_is_pcm16, rng_pos);
	if (rc) return rc;
	if (!coded_sp && !coded_ap) return WC_OK;
	WC_HIP(hipSetDevice(p->dev->id));
	DeviceLock lock(p->dev);
	hipStream_t s = p->dev->active();
	const int n_apc = GetNumberOfAperiodicities(p->fs);
	std::vector<int> f_len(n_utt);
	long long nf = 0;
	for (int u = 0; u < n_utt; ++u) { f_len[u] = wc_get_samples(p->fs, x_length[u], p->frame_period); nf += f_len[u]; }
	const size_t b_sp = coded_sp ? sizeof(double) * (size_t)nf * number_of_dimensions : 0, b_ap = coded_ap ? sizeof(double) * (size_t)nf * n_apc : 0;
	if ((rc = p->b_coded.reserve(b_sp + b_ap + 16))) return rc;
	if ((rc = p->st_out.reserve(b_sp + b_ap + 16))) return rc;
	double *d_csp = p->b_coded.as<double>(), *d_cap = d_csp + (b_sp / sizeof(double));
	if (coded_sp && (rc = wc_code_spectral_envelope_device(p->fs, p->fft_size, nf, number_of_dimensions, p->b_sp.as<double>(), d_csp))) return rc;
	if (coded_ap && (rc = wc_code_aperiodicity_device(p->fs, p->fft_size, nf, p->b_ap.as<double>(), d_cap))) return rc;
	char *out = static_cast<char *>(p->st_out.p);
	if (b_sp + b_ap) WC_HIP(hipMemcpyAsync(out, d_csp, b_sp + b_ap, hipMemcpyDeviceToHost, s));
	WC_HIP(hipStreamSynchronize(s));
	std::vector<CopyJob> jobs;
	long long fo = 0;
	for (int u = 0; u < n_utt; ++u) {
		if (coded_sp && coded_sp[u]) jobs.push_back({coded_sp[u], out + sizeof(double) * (size_t)fo * number_of_dimensions, sizeof(double) * (size_t)f_len[u] * number_of_dimensions});
		if (coded_ap && coded_ap[u]) jobs.push_back({coded_ap[u], out + b_sp + sizeof(double) * (size_t)fo * n_apc, sizeof(double) * (size_t)f_len[u] * n_apc});
		fo += f_len[u];
	}
	parallel_copy(jobs);
	return WC_OK;
}


}  // extern "C"
