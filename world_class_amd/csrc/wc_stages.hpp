// Enqueue-only entry points of the four stages (no host synchronisation), used by the fused pipeline
// (wc_pipeline.hip).  Each is defined next to its stage's kernels.
#pragma once
#include <utility>
#include <vector>

#include "wc_internal.hpp"

// part: 3 = the whole chain, 1 = its front only, 2 = the tail behind a front that an earlier call with the same arguments enqueued;
// bp_done: recorded behind the band-pass; tail_after: the tail waits for it
int hv_enqueue(wc_harvest *h, hipStream_t s, int n_utt, const double *d_x, const int *x_length, double *d_tpos, double *d_f0,
			   bool full, hipEvent_t mid_event, hipEvent_t start_after, int part = 3, hipEvent_t bp_done = nullptr,
			   hipEvent_t tail_after = nullptr);
int hv_overflowed(wc_harvest *h, hipStream_t s, bool *overflow, bool *tie = nullptr, std::vector<int> *tie_utts = nullptr);
// stretches [u0, u1) of consecutive utterances out of a sorted list
std::vector<std::pair<int, int>> hv_runs_of(const std::vector<int> &us);
// the utterances whose refinement raised the tie flag are run again on this handle: the same options, band-pass as direct FIR sums
wc_harvest *hv_exact_twin(wc_harvest *h);
// Harvest in two parts (incremental streams): phases 1 = front (decimation .. refinement), 2 = tail (unreliable .. output), 3 = both
void hv_set_phases(wc_harvest *h, int mask);
int hv_row_width(const wc_harvest *h);
int hv_reserve_rows(wc_harvest *h, long long total_1ms_frames);
double *hv_candidate_rows(wc_harvest *h);
double *hv_score_rows(wc_harvest *h);

int ct_prepare(wc_cheaptrick *c, hipStream_t s, int n_utt, const int *x_length, const double *d_f0, const int *f0_length,
			   const uint64_t *rng_pos, long long *total_out, uint64_t *min_pos_out, uint64_t *max_end_out);
int ct_frames(wc_cheaptrick *c, hipStream_t s, int n_utt, const double *d_x, const double *d_tpos, const double *d_f0,
			  double *d_sp, long long total, hipEvent_t *rows_done);
const unsigned long long *ct_end_positions(const wc_cheaptrick *c);
// the caller vouches that no F0 handed to the stage exceeds f0_bound (0: no promise): passes over frames with F0 in the kHz are not launched
void ct_set_f0_bound(wc_cheaptrick *c, double f0_bound);
void d4c_set_f0_bound(wc_d4c *d, double f0_bound);

int d4c_enqueue(wc_d4c *d, hipStream_t s, int n_utt, const double *d_x, const int *x_length, const double *d_tpos,
				const double *d_f0, const int *f0_length, int fft_size, double *d_ap, const uint64_t *rng_pos,
				const unsigned long long *d_start);
const unsigned long long *d4c_end_positions(const wc_d4c *d);
uint64_t d4c_draw_bound(const wc_d4c *d, int f0_length);

int syn_prepare(wc_synthesis *sy, hipStream_t s, int n_utt, const double *d_f0, const int *f0_length, const int *out_length,
				double *d_out, const uint64_t *rng_pos, bool full);
int syn_pulses(wc_synthesis *sy, hipStream_t s, const double *d_f0, const double *d_sp, const double *d_ap, double *d_out,
			   const unsigned long long *d_start);
int syn_finish(wc_synthesis *sy, hipStream_t s, uint64_t *rng_pos_out, bool *overflow);

// the device a handle lives on (the host-pointer batch entry points stage their buffers there, not on the calling thread's device)
wc::Device *hv_device(const wc_harvest *h);
wc::Device *ct_device(const wc_cheaptrick *c);
wc::Device *d4c_device(const wc_d4c *d);
wc::Device *syn_device(const wc_synthesis *sy);
