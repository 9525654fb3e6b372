// Harvest F0 estimation on gfx950.
//
// Restates reference src/harvest.cpp:183-1453 (compute, generalBody and everything they reach) and
// decimate/FilterForDecimate of src/world_matlabfunctions.cpp:27-125, :184-210 as a chain of kernels:
//
//   hv_decimate_lds_kernel  zero-phase order-3 IIR decimation (reference :213-248).  The recursion is
//                        split into independent chunks that warm up on the preceding samples (the
//                        filter's impulse response is below 1e-20 after 768 samples), two passes; the
//                        lanes' streams are staged through LDS in tiles (hv_decimate_kernel reads them
//                        directly, WC_HARVEST_DECIMATE=direct)
//   hv_dc_kernel         the reference's int-typed "DC removal" (:239) -- a no-op unless abs(y) >= 1
//   hv_bandpass_sdft_kernel, hv_compact_kernel
//                        the band-pass of reference :1261-1305 -- a Nuttall * cosine FIR of <= 2*1024+1 taps -- as a
//                        sliding DFT (a lane per (band, 2048-sample chunk), seven rotating sums) fused with the four
//                        zero-crossing detectors of :1179-1255; edges land in per-chunk slots that the second kernel
//                        packs in time order.  hv_bandpass_kernel is the direct FIR evaluation (one workgroup per
//                        (utterance, band), register tiled from LDS), kept behind WC_HARVEST_BANDPASS=fir
//   hv_raw_kernel        interp1 of the four interval series onto the 1 ms grid (:1098-1143)
//   hv_detect_kernel     per-frame candidate detection over bands (:1005-1083)
//   hv_refine_packed_kernel  eight lanes per (frame, candidate): overlap (:987-1000) folded into the
//                        gather, Blackman / differentiated windows (<true>: from the reference's cosine
//                        table, HarvestOption::use_cos_table), and -- instead of the reference's two
//                        full FFTs -- Goertzel recurrences for the <= 6 harmonic bins fixF0 reads (:809-927).
//                        One wavefront per frame: live candidates packed eight to a pass, candidates that share
//                        window length and bins computed once.  hv_refine_kernel is the plain layout (a wavefront
//                        per candidate slot, the seven overlap blocks side by side), kept behind WC_HARVEST_REFINE=slots
//   hv_unreliable_kernel (:708-744)
//   hv_contour_kernel<0/1/2>  the sequential contour logic (:254-634: fixStep1..4, extend, merge): one
//                        wavefront per utterance with the candidate searches spread over the lanes, except
//                        the extension walks (<1>), which get a wavefront per (voiced section, direction)
//   hv_smooth_kernel     zero-lag Butterworth per voiced section, one lane per section (:639-703)
//   hv_output_kernel     1 ms contour -> frame_period grid (:199-204)
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "wc_argsort.hpp"
#include <algorithm>

#include "wc_device.hpp"
#include "wc_internal.hpp"
#include "wc_frames.hpp"
#include "wc_hostcopy.hpp"

namespace wc {

constexpr double kSafeH = 0.000000000001;
constexpr int HL_MAX = 1024;      // longest supported half filter length (f0_floor >= ~18 Hz at 8 kHz after decimation, ~35 Hz at 16 kHz)
constexpr int BP_T = 256;         // threads of the band-pass workgroup
constexpr int BP_R = 8;           // consecutive outputs per thread
constexpr int BP_TILE = BP_T * BP_R;
constexpr int BP_ADV = BP_TILE;      // tile unit of the per-tile edge counts (tile_run) that hv_raw reads
constexpr int SD_CH = BP_ADV;       // output samples per lane of the sliding band-pass
constexpr int FIR_ADV = BP_TILE - 2; // tile advance of the FIR band-pass: its detectors look two samples ahead
// zero margins around every utterance's decimated signal, so the sliding band-pass reads without bounds checks
constexpr int Y_PADL = 2 * HL_MAX + 16, Y_PADR = SD_CH + 2 * HL_MAX + 16;
constexpr int MAX_SLOTS = 32;     // candidates per frame before overlap (reference: round(bands/10))

// orders a wavefront's own LDS writes before its later reads (its LDS operations execute in order: this only keeps the compiler from
// moving them; no wait, no s_barrier)
__device__ __forceinline__ void rq_fence() {
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#ifndef WC_RQ_PROF
#define WC_RQ_PROF 0  // development builds: shader-clock cycles of every wavefront by phase, summed over the launch (printed by launch_refine)
#endif
#if WC_RQ_PROF
__device__ unsigned long long rq_prof[16];
#define RQ_T(k) do { const long long now_ = clock64(); acc_[k] += now_ - last_; last_ = now_; } while (0)
#define RQ_FLUSH() do { if (lane == 0 && ((blockIdx.x + 7 * blockIdx.y) & 63) == 0) for (int k_ = 0; k_ < 12; ++k_) atomicAdd(&rq_prof[k_], (unsigned long long)acc_[k_]); } while (0)
#else
#define RQ_T(k) do {} while (0)
#define RQ_FLUSH() do {} while (0)
#endif

struct HvUtt {
	long long x_off;     // samples
	long long dec_off;   // scratch of the decimator (x_len + 2 lag + 18 per utterance)
	long long y_off;     // decimated signal
	long long l1_off;    // 1 ms frames
	long long out_off;   // output frames
	long long ev_off;    // first event slot of this utterance
	int x_len, y_len, L1, L;
};

struct HvParams {
	int fs, decim, n_bands, S, n_cand;  // S = slots per overlap block, n_cand = 7 S
	double fs_d, f0_floor, f0_ceil, frame_period;
};

__device__ __forceinline__ int hv_find(const HvUtt *__restrict__ u, int n, long long v, long long HvUtt::*field) {
	int lo = 0, hi = n - 1;
	while (lo < hi) {
		int mid = (lo + hi + 1) >> 1;
		if (u[mid].*field <= v) lo = mid; else hi = mid - 1;
	}
	return lo;
}

// ------------------------------------------------------------------------------------------------
// decimation
// ------------------------------------------------------------------------------------------------
struct DecCoef { double a0, a1, a2, b0, b1; };
#ifndef WC_DEC_CHUNK
#define WC_DEC_CHUNK 512
#endif
constexpr int DEC_CHUNK = WC_DEC_CHUNK, DEC_WARM = 768;

// pass 0: forward over the edge-padded input; pass 1: forward over the reversed pass-0 output, storing
// only the samples the decimated signal keeps.  Index algebra of reference
// src/world_matlabfunctions.cpp:184-210 and src/harvest.cpp:222-235.
template <int PASS>
__global__ void hv_decimate_kernel(const HvUtt *__restrict__ utts, const double *__restrict__ x, double *__restrict__ buf,
								   double *__restrict__ y, DecCoef c, int r, int lag) {
	const HvUtt u = utts[blockIdx.y];
	const int nn = u.x_len + 2 * lag;
	const int len = nn + 18;
	const int chunk = blockIdx.x * blockDim.x + threadIdx.x;
	const int k0 = chunk * DEC_CHUNK;
	if (k0 >= len) return;
	const int k1 = min(len, k0 + DEC_CHUNK);
	const int ks = max(0, k0 - DEC_WARM);
	const double *__restrict__ xin = x + u.x_off;
	double *__restrict__ b = buf + u.dec_off;
	double w0 = 0.0, w1 = 0.0, w2 = 0.0;
	// output bookkeeping of pass 1
	const int nout = nn / r + 1;
	const int nbeg = r - r * nout + nn;
	const int first = nbeg + (lag / r) * r + 8;  // index (original order) of y[0]
#pragma unroll 4
	for (int k = ks; k < k1; ++k) {
		double v;
		if (PASS == 0) v = xin[clampi(k - 9 - lag, 0, u.x_len - 1)];
		else v = b[len - 1 - k];
		double wt = v + c.a0 * w0 + c.a1 * w1 + c.a2 * w2;
		double o = c.b0 * wt + c.b1 * w0 + c.b1 * w1 + c.b0 * w2;
		w2 = w1; w1 = w0; w0 = wt;
		if (k >= k0) {
			if (PASS == 0) {
				b[k] = o;
			} else {
				int p = len - 1 - k;  // position in the original order
				int d = p - first;
				if (d >= 0 && d % r == 0) {
					int i = d / r;
					if (i < u.y_len) y[u.y_off + i] = o;
				}
			}
		}
	}
}

// The same two passes with the samples staged through LDS.  A lane walks its own chunk, so the direct kernel above reads 64
// streams 8 KB apart with every load instruction -- 64 pages and 64 cache lines for 512 bytes -- and spends its time in
// address translation and miss latency (0.45 ms per pass for 32 x 10 s at 48 kHz, 7x the time the bytes need).  Here the
// wavefront fetches a tile of DEC_TS samples of every lane's stream with loads that cover two whole rows (2 x 256 contiguous
// bytes) each, one tile ahead of the arithmetic, and pass 0 returns its outputs the same way.  The recursion, its operation
// order and the chunk / warm-up boundaries are those of the direct kernel, so the output is bit-identical: a lane's
// range is the uniform [k0 - DEC_WARM, k0 + DEC_CHUNK); positions before the signal feed zeros into a zero state, which
// stays exactly zero, and positions past its end are computed and dropped.
constexpr int DEC_TS = 32;
template <int PASS>
__global__ __launch_bounds__(64) void hv_decimate_lds_kernel(const HvUtt *__restrict__ utts, const double *__restrict__ x,
															  double *__restrict__ buf, double *__restrict__ y, DecCoef c, int r, int lag) {
	__shared__ double tile[64][DEC_TS + 1];
	const HvUtt u = utts[blockIdx.y];
	const int nn = u.x_len + 2 * lag;
	const int len = nn + 18;
	const int lane = threadIdx.x;
	const int chunk0 = blockIdx.x * 64;
	if (chunk0 * DEC_CHUNK >= len) return;
	const double *__restrict__ xin = x + u.x_off;
	double *__restrict__ b = buf + u.dec_off;
	const int nout = nn / r + 1;
	const int nbeg = r - r * nout + nn;
	const int first = nbeg + (lag / r) * r + 8;  // index (original order) of y[0]
	const int k0 = (chunk0 + lane) * DEC_CHUNK;
	const int jj = lane & (DEC_TS - 1), rh = lane >> 5;  // this lane's column and row parity in the cooperative transfers
	auto fetch = [&](int t, double (&v)[DEC_TS]) {
#pragma unroll
		for (int i = 0; i < DEC_TS; ++i) {
			const int row = 2 * i + rh;
			const int k = (chunk0 + row) * DEC_CHUNK - DEC_WARM + t * DEC_TS + jj;
			double q = 0.0;
			if (k >= 0) {
				if (PASS == 0) q = xin[clampi(k - 9 - lag, 0, u.x_len - 1)];
				else if (k < len) q = b[len - 1 - k];
			}
			v[i] = q;
		}
	};
	constexpr int NT = (DEC_WARM + DEC_CHUNK) / DEC_TS;
	static_assert((DEC_WARM + DEC_CHUNK) % DEC_TS == 0 && DEC_WARM % DEC_TS == 0, "tiles must not straddle the warm-up boundary");
	double nxt[DEC_TS];
	fetch(0, nxt);
	double w0 = 0.0, w1 = 0.0, w2 = 0.0;
	for (int t = 0; t < NT; ++t) {
		__syncthreads();  // the previous tile has been consumed (and, in pass 0, its outputs collected)
#pragma unroll
		for (int i = 0; i < DEC_TS; ++i) tile[2 * i + rh][jj] = nxt[i];
		__syncthreads();
		if (t + 1 < NT) fetch(t + 1, nxt);
		const int kt = k0 - DEC_WARM + t * DEC_TS;
		const bool keep = t * DEC_TS >= DEC_WARM;  // wave-uniform: beyond the warm-up
#pragma unroll
		for (int j = 0; j < DEC_TS; ++j) {
			const double v = tile[lane][j];
			double wt = v + c.a0 * w0 + c.a1 * w1 + c.a2 * w2;
			double o = c.b0 * wt + c.b1 * w0 + c.b1 * w1 + c.b0 * w2;
			w2 = w1; w1 = w0; w0 = wt;
			if (PASS == 0) {
				if (keep) tile[lane][j] = o;  // own row, own column: nobody else touches it before the barrier
			} else if (keep) {
				const int k = kt + j;
				if (k < len) {
					const int p = len - 1 - k;  // position in the original order
					const int d = p - first;
					if (d >= 0 && d % r == 0) {
						const int i = d / r;
						if (i < u.y_len) y[u.y_off + i] = o;
					}
				}
			}
		}
		if (PASS == 0 && keep) {
			__syncthreads();
#pragma unroll
			for (int i = 0; i < DEC_TS; ++i) {
				const int row = 2 * i + rh;
				const int k = (chunk0 + row) * DEC_CHUNK - DEC_WARM + t * DEC_TS + jj;
				if (k < len) b[k] = tile[row][jj];
			}
		}
	}
}

// Round 6 (default): the same two passes with EXACT chunk states instead of warm-ups.  The recursion is linear: the state behind a
// chunk is A^C times the state in front of it plus the state the chunk's own samples leave behind a zero state.  A thread
// holds the DS_C samples of its chunk in registers, runs them through the recursion from a zero state, the workgroup scans those
// end states with the constant matrices A^(C 2^j) (seven levels, no sample touched), and every thread runs its chunk once more
// from the state the scan hands it -- the reference's statements in the reference's order, on a starting state that is
// exact up to rounding (the chunked kernels above start 768 samples early instead and re-read those: 2.5 samples fetched
// and stepped per sample kept at 512-sample chunks, far more at the small chunks a lone utterance would need to fill the
// chip).  Only a workgroup's first DS_WARM threads are warm-up (their chunks start from a zero state, as the kernels above do
// everywhere): 5 % more samples, and exactly zero at the start of an utterance.  Chunks of 32 samples: an utterance of 10 s is
// 250 wavefronts per pass where the kernels above have 15.  Results within an ulp of the chunked kernels' (a starting state
// rounds differently), not the same bits: tests/test_gpu_harvest.py holds the two to 1e-14 of the signal's scale.
constexpr int DS_C = 32, DS_T = 512, DS_WARM = DEC_WARM / DS_C, DS_SPAN = (DS_T - DS_WARM) * DS_C;
struct DecScan { double mp[7][9]; };  // A^(DS_C 2^j), row-major
__device__ __forceinline__ void ds_madd(double (&s)[3], const double (&m)[9], const double (&t)[3]) {  // s += m t
	s[0] += m[0] * t[0] + m[1] * t[1] + m[2] * t[2];
	s[1] += m[3] * t[0] + m[4] * t[1] + m[5] * t[2];
	s[2] += m[6] * t[0] + m[7] * t[1] + m[8] * t[2];
}
template <int PASS>
__global__ __launch_bounds__(DS_T) void hv_decimate_scan_kernel(const HvUtt *__restrict__ utts, const double *__restrict__ x, double *__restrict__ buf,
																 double *__restrict__ y, DecCoef c, DecScan ms, const double *__restrict__ ptab, int r, int lag) {
	__shared__ double tot[DS_T / 64][3];
	__shared__ double tiles[DS_T / 64][32 * 33];  // per wavefront: the chunks of 32 of its lanes on their way between memory order and lane order
	const HvUtt u = utts[blockIdx.y];
	const int nn = u.x_len + 2 * lag;
	const int len = nn + 18;
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const long long span0 = (long long)blockIdx.x * DS_SPAN;
	if (span0 >= len) return;
	const double *__restrict__ xin = x + u.x_off;
	double *__restrict__ b = buf + u.dec_off;
	const int k0 = (int)span0 + (tid - DS_WARM) * DS_C;  // (negative in front of the signal: zeros into a zero state)
	// A wavefront's 64 chunks are 2048 consecutive positions: fetched 64 consecutive ones per instruction (a lane reading its own
	// chunk touches 64 cache lines per load) and handed to their lanes through LDS, the chunks of 32 lanes at a time
	double *const tile = &tiles[wv][0];
	const int kw0 = (int)span0 + (wv * 64 - DS_WARM) * DS_C;
	double v[DS_C];
#pragma unroll
	for (int hf = 0; hf < 2; ++hf) {
		double q[16];
#pragma unroll
		for (int i = 0; i < 16; ++i) {
			const int k = kw0 + 1024 * hf + lane + 64 * i;
			q[i] = 0.0;
			if (k >= 0 && k < len) q[i] = (PASS == 0) ? xin[clampi(k - 9 - lag, 0, u.x_len - 1)] : b[len - 1 - k];
		}
#pragma unroll
		for (int i = 0; i < 16; ++i) {
			const int e = lane + 64 * i;
			tile[(e >> 5) * 33 + (e & 31)] = q[i];
		}
		rq_fence();
		if ((lane >> 5) == hf) {
#pragma unroll
			for (int j = 0; j < DS_C; ++j) v[j] = tile[(lane & 31) * 33 + j];
		}
		rq_fence();
	}
	// the state the chunk leaves behind a zero state
	double st[3] = {0.0, 0.0, 0.0};
#pragma unroll
	for (int j = 0; j < DS_C; ++j) {
		const double wt = v[j] + c.a0 * st[0] + c.a1 * st[1] + c.a2 * st[2];
		st[2] = st[1]; st[1] = st[0]; st[0] = wt;
	}
	// ... scanned: first inside the wavefront,
#pragma unroll
	for (int j = 0; j < 6; ++j) {
		double t[3];
#pragma unroll
		for (int k = 0; k < 3; ++k) t[k] = __shfl_up(st[k], 1 << j, 64);
		if (lane >= (1 << j)) ds_madd(st, ms.mp[j], t);
	}
	if (lane == 63) { tot[wv][0] = st[0]; tot[wv][1] = st[1]; tot[wv][2] = st[2]; }
	__syncthreads();
	// then the state in front of this wavefront from the totals of the ones before it,
	double pre[3] = {0.0, 0.0, 0.0};
	for (int w = 0; w < wv; ++w) {
		double nx[3] = {tot[w][0], tot[w][1], tot[w][2]};
		ds_madd(nx, ms.mp[6], pre);
		pre[0] = nx[0]; pre[1] = nx[1]; pre[2] = nx[2];
	}
	// carried to every chunk's end (A^(C (lane + 1)) from a table), and the state in FRONT of the chunk is its neighbour's
	{
		double m[9];
#pragma unroll
		for (int k = 0; k < 9; ++k) m[k] = ptab[9 * lane + k];
		ds_madd(st, m, pre);
	}
	double w0 = __shfl_up(st[0], 1, 64), w1 = __shfl_up(st[1], 1, 64), w2 = __shfl_up(st[2], 1, 64);
	if (lane == 0) { w0 = pre[0]; w1 = pre[1]; w2 = pre[2]; }
	// the chunk once more, from its own starting state (reference src/world_matlabfunctions.cpp:106-118, statement by statement)
	const int nout = nn / r + 1;
	const int nbeg = r - r * nout + nn;
	const int first = nbeg + (lag / r) * r + 8;  // index (original order) of y[0]
	const bool keep = tid >= DS_WARM;
#pragma unroll
	for (int j = 0; j < DS_C; ++j) {
		const double wt = v[j] + c.a0 * w0 + c.a1 * w1 + c.a2 * w2;
		const double o = c.b0 * wt + c.b1 * w0 + c.b1 * w1 + c.b0 * w2;
		w2 = w1; w1 = w0; w0 = wt;
		const int k = k0 + j;
		if (PASS == 0) {
			v[j] = o;  // (leaves through LDS below)
		} else if (keep && k < len) {
			const int p = len - 1 - k;  // position in the original order
			const int d = p - first;
			if (d >= 0 && d % r == 0) {
				const int i = d / r;
				// (The recursion's tail behind a stretch that ends in digital silence is carried exactly here -- the chunked kernels above cut
				// it off 768 samples on, where a chunk's warm-up starts from zero -- and would go on down into the denormal numbers, where the
				// sliding band-pass's reciprocal-based quotients of neighbouring outputs have no answer: 240 orders of magnitude below
				// full scale the tail is set to the zero it becomes 0.1 s later anyway.)
				if (i < u.y_len) y[u.y_off + i] = fabs(o) < 0x1p-800 ? 0.0 : o;
			}
		}
	}
	if (PASS == 0) {
#pragma unroll
		for (int hf = 0; hf < 2; ++hf) {
			if ((lane >> 5) == hf) {
#pragma unroll
				for (int j = 0; j < DS_C; ++j) tile[(lane & 31) * 33 + j] = v[j];
			}
			rq_fence();
#pragma unroll
			for (int i = 0; i < 16; ++i) {
				const int e = lane + 64 * i;
				const int k = kw0 + 1024 * hf + e;
				if (k >= (int)span0 && k < len) b[k] = tile[(e >> 5) * 33 + (e & 31)];
			}
			rq_fence();
		}
	}
}

// reference src/harvest.cpp:237-241: accumulate(y, y + n, 0) with an int accumulator truncates after every
// addition, so the "mean" is 0 unless some abs(y) reaches 1.  Emulated exactly.
__global__ void hv_dc_kernel(const HvUtt *__restrict__ utts, double *__restrict__ y) {
	__shared__ int flag;
	__shared__ double mean;
	const HvUtt u = utts[blockIdx.x];
	double *__restrict__ yy = y + u.y_off;
	if (threadIdx.x == 0) flag = 0;
	__syncthreads();
	int f = 0;
	for (int i0 = threadIdx.x; i0 < u.y_len; i0 += 8 * blockDim.x) {  // eight loads in flight per thread: this scan heads every batch
		double v[8];
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			const int i = i0 + k * blockDim.x;
			v[k] = i < u.y_len ? yy[i] : 0.0;
		}
#pragma unroll
		for (int k = 0; k < 8; ++k) if (fabs(v[k]) >= 1.0) f = 1;
	}
	if (f) flag = 1;
	__syncthreads();
	if (!flag) return;
	if (threadIdx.x == 0) {
		int acc = 0;
		for (int i = 0; i < u.y_len; ++i) acc = (int)(acc + yy[i]);
		double m = acc;
		mean = m / u.y_len;
	}
	__syncthreads();
	for (int i = threadIdx.x; i < u.y_len; i += blockDim.x) yy[i] -= mean;
}
__global__ void hv_copy_kernel(const HvUtt *__restrict__ utts, const double *__restrict__ x, double *__restrict__ y) {
	const HvUtt u = utts[blockIdx.y];
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < u.y_len) y[u.y_off + i] = (i < u.x_len) ? x[u.x_off + i] : 0.0;  // decimation_ratio == 1 (reference :217-219)
}

// ------------------------------------------------------------------------------------------------
// band-pass + four zero-crossing detectors
// ------------------------------------------------------------------------------------------------
struct BpArgs {
	const HvUtt *utts;
	const double *y;
	const double *taps;       // per band, padded to a multiple of 8 with zeros
	const int *tap_off;       // first tap of each band
	const int *half_len;      // hl per band
	const long long *ev_band_off;  // per band: first slot relative to the utterance's ev_off (4 types contiguous)
	const int *ev_cap;        // per band capacity per type
	double *events;           // fine edge positions
	int *ev_count;            // [utt][band][4]
	int *overflow;
	int *tile_run;   // [utt][band][n_tiles + 1][4]: edges found before each tile (lets hv_raw bound its slices)
	int n_tiles;
	int n_bands;
};

__device__ __forceinline__ int padidx(int m) { return m + (m >> 3); }

__global__ __launch_bounds__(BP_T, 3) void hv_bandpass_kernel(BpArgs a) {
	// 36.5 KB of LDS -> four workgroups per CU; the filtered tile reuses the signal tile's storage
	__shared__ double Ys[(BP_TILE + 2 * HL_MAX + 32) * 9 / 8 + 16];
	__shared__ double Tp[2 * HL_MAX + 16];
	double *Ss = Ys;
	__shared__ unsigned long long scan_s[BP_T / 64];
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int band = blockIdx.x;
	const HvUtt u = a.utts[blockIdx.y];
	const int hl = a.half_len[band];
	const int nt8 = ((2 * hl + 1 + 7) / 8) * 8;
	const double *__restrict__ y = a.y + u.y_off;
	const double *__restrict__ taps = a.taps + a.tap_off[band];
	for (int i = tid; i < nt8; i += BP_T) Tp[i] = taps[i];
	const int cap = a.ev_cap[band];
	double *__restrict__ ev = a.events + u.ev_off + a.ev_band_off[band];
	int run[4] = {0, 0, 0, 0};
	const int t0 = tid * BP_R;
	int *__restrict__ trun = a.tile_run + ((long long)blockIdx.y * a.n_bands + band) * (a.n_tiles + 1) * 4;
	for (int ts = 0; ts < u.y_len; ts += FIR_ADV) {
		if (tid < 4) trun[(ts / FIR_ADV) * 4 + tid] = run[tid];
		// Ys[m] = y[ts + 1 - hl + m], m in [0, TILE + nt8 + 8)
		__syncthreads();
		for (int m = tid; m < BP_TILE + nt8 + 8; m += BP_T) {
			int g = ts + 1 - hl + m;
			Ys[padidx(m)] = (g >= 0 && g < u.y_len) ? y[g] : 0.0;
		}
		__syncthreads();
		// out[t0 + j] = sum_q tap[q] * Ys[t0 + j + q]  (the filter is symmetric)
		double acc[BP_R];
#pragma unroll
		for (int j = 0; j < BP_R; ++j) acc[j] = 0.0;
		double w[16];
#pragma unroll
		for (int c = 0; c < 8; ++c) w[c] = Ys[padidx(t0 + c)];
		for (int q0 = 0; q0 < nt8; q0 += 8) {
			double tp[8];
#pragma unroll
			for (int c = 0; c < 8; ++c) { w[8 + c] = Ys[padidx(t0 + q0 + 8 + c)]; tp[c] = Tp[q0 + c]; }
#pragma unroll
			for (int qq = 0; qq < 8; ++qq)
#pragma unroll
				for (int j = 0; j < BP_R; ++j) acc[j] = fma(tp[qq], w[qq + j], acc[j]);
#pragma unroll
			for (int c = 0; c < 8; ++c) w[c] = w[8 + c];
		}
		__syncthreads();  // every thread is done reading the signal tile
#pragma unroll
		for (int j = 0; j < BP_R; ++j) Ss[padidx(t0 + j)] = acc[j];
		__syncthreads();
		// detectors at positions i = ts + t, t in [0, BP_ADV)
		double s[BP_R + 2];
#pragma unroll
		for (int j = 0; j < BP_R + 2; ++j) s[j] = (t0 + j < BP_TILE) ? Ss[padidx(t0 + j)] : 0.0;
		unsigned int mask[4] = {0, 0, 0, 0};
#pragma unroll
		for (int j = 0; j < BP_R; ++j) {
			const int t = t0 + j, i = ts + t;
			if (t < FIR_ADV) {
				const double s0 = s[j], s1 = s[j + 1], s2 = s[j + 2];
				if (i + 1 < u.y_len) {  // zeroCrossingEngine over y_length samples
					if (0.0 < s0 && s1 <= 0.0) mask[0] |= 1u << j;    // positive -> negative
					if (0.0 < -s0 && -s1 <= 0.0) mask[1] |= 1u << j;  // negative -> positive
				}
				if (i + 2 < u.y_len) {  // engine over the y_length - 1 first differences
					const double d0 = s1 - s0, d1 = s2 - s1;  // = (-s[i]) - (-s[i+1])
					if (0.0 < d0 && d1 <= 0.0) mask[2] |= 1u << j;    // peaks
					if (0.0 < -d0 && -d1 <= 0.0) mask[3] |= 1u << j;  // dips
				}
			}
		}
		// ordered compaction: the four counts travel packed in one 64-bit word
		unsigned long long packed = 0;
#pragma unroll
		for (int ty = 0; ty < 4; ++ty) packed |= (unsigned long long)__popc(mask[ty]) << (16 * ty);
		unsigned long long inc = packed;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			unsigned long long t = __shfl_up(inc, o, 64);
			if (lane >= o) inc += t;
		}
		if (lane == 63) scan_s[wv] = inc;
		__syncthreads();
		unsigned long long base = 0, total = 0;
#pragma unroll
		for (int k = 0; k < BP_T / 64; ++k) {
			unsigned long long t = scan_s[k];
			if (k < wv) base += t;
			total += t;
		}
		const unsigned long long excl = base + inc - packed;
#pragma unroll
		for (int ty = 0; ty < 4; ++ty) {
			int slot = run[ty] + (int)((excl >> (16 * ty)) & 0xffffull);
#pragma unroll
			for (int j = 0; j < BP_R; ++j) {
				if (mask[ty] & (1u << j)) {
					const int i = ts + t0 + j;
					double v0, v1;
					if (ty < 2) { v0 = s[j]; v1 = s[j + 1]; }
					else { v0 = s[j + 1] - s[j]; v1 = s[j + 2] - s[j + 1]; }
					// edges[e] - sig[e-1] / (sig[e] - sig[e-1]) with edges = i + 1 (reference :1206-1208);
					// the value is the same for a signal and its negation
					const double fine = (i + 1) - v0 / (v1 - v0);
					if (slot < cap) ev[(long long)ty * cap + slot] = fine;
					++slot;
				}
			}
			run[ty] += (int)((total >> (16 * ty)) & 0xffffull);
		}
	}
	if (tid < 4) {
		for (int q = (u.y_len + FIR_ADV - 1) / FIR_ADV; q <= a.n_tiles; ++q) trun[q * 4 + tid] = run[tid];
		int cnt = run[tid];
		a.ev_count[((long long)blockIdx.y * a.n_bands + band) * 4 + tid] = cnt;
		if (cnt > cap) atomicExch(a.overflow, 1);
	}
}

// ------------------------------------------------------------------------------------------------
// The same band-pass as a sliding DFT.  The filter is a Nuttall window (a sum of four cosines of k Omega,
// Omega = pi / hl) times cos(w (k - hl)), so its output is a weighted sum of the real parts of seven sliding
// sums  D_v(m) = P0 sum_{k=0}^{2hl} y[m-k] e^{i v k},  v = w, w +- Omega, w +- 2 Omega, w +- 3 Omega,
// P0 = e^{-i w hl}, with weights a0, -a1/2, -a1/2, a2/2, a2/2, -a3/2, -a3/2.  Each obeys
//      D_v(m+1) = R_v (D_v(m) - y[m-2hl] conj(P0)) + y[m+1] P0,      R_v = e^{i v},
// (e^{i v 2hl} is the same for all seven v), i.e. 6 instructions per sample and frequency instead of 2 hl + 1
// multiply-adds per sample.  A lane owns one (band, chunk of SD_CH output samples): it builds its sums from
// zero over the first window (the same recurrence without the leaving sample), then slides, running the four
// zero-crossing detectors of reference :1179-1255 on its outputs as they appear (rounding errors of the
// rotations random-walk over one chunk only: ~1e-14 relative).  Edges go to per-(band, chunk, type) slots;
// hv_compact_kernel fills tile_run (edges before every chunk) and, for WC_HARVEST_RAW=lists only, packs the edges in time order into per-band lists.
// ------------------------------------------------------------------------------------------------
struct SdArgs {
	const HvUtt *utts;
	const double *y;
	const double2 *rot;        // [band][7]  R_v
	const double2 *p0;         // [band]     P0
	const int *half_len;       // hl per band
	const long long *slot_off; // per band: first slot relative to the utterance's slot base
	const int *slot_cap;       // per band: capacity of one (chunk, type) slot
	long long slots_per_utt;
	double *slots;
	int *slot_count;           // [utt][band][chunk][4]
	int n_bands, n_chunks;
	const double *taps;        // the filters themselves (per band, as the FIR band-pass reads them) ...
	const int *tap_off;
	double *seam;              // ... and what hv_seam_kernel makes of them: [utt][band][n_chunks + 1][2], the first two outputs of every chunk
	const int *quiet;          // [utt][n_chunks]: chunks the sliding sums leave to hv_bandpass_quiet_kernel (hv_quiet_kernel), then [utt]: any
	const double *bmax;        // [utt][n_blk]: largest |y| of every 64 samples (hv_blockmax_kernel)
	int n_blk, hl_max;
};

// Seams (round 5).  A chunk's lane and its neighbour's both need the two outputs on either side of their common border -- each
// detector looks two samples ahead -- and each would compute them with its own recurrence: two estimates of one sample, 1e-14 of
// the signal apart.  Where the true value is zero (a train of impulses leaves exact zeros and exactly symmetric extrema between
// its pulses) one lane then sees a crossing at the last sample of its chunk and the other one at the first sample of the next:
// the same edge twice, an interval of nothing, a NaN in the raw candidates of 41 bands around every border
// (profiles/r05_c_impulse_trains.txt: three voicing flips against the reference on one train in forty).  The first two outputs of
// every chunk are therefore computed ONCE, here, as direct sums of the filter itself (1e-16, like the reference's FFT
// convolution), and both lanes use these: every sample has one value again.  One thread per (band, chunk), the band's taps in
// LDS; 0.02 ms per 64 x 10 s.
__global__ __launch_bounds__(64) void hv_seam_kernel(SdArgs a) {
	__shared__ double Tp[2 * HL_MAX + 16];
	const int band = blockIdx.x;
	const HvUtt u = a.utts[blockIdx.y];
	const int hl = a.half_len[band], nt = 2 * hl + 1;
	const double *__restrict__ taps = a.taps + a.tap_off[band];
	for (int q = threadIdx.x; q < nt; q += 64) Tp[q] = taps[q];
	__syncthreads();
	const double *__restrict__ y = a.y + u.y_off;  // (zero margins of Y_PADL / Y_PADR samples)
	double *__restrict__ out = a.seam + ((long long)blockIdx.y * a.n_bands + band) * (a.n_chunks + 1) * 2;
	for (int c = threadIdx.x; c <= a.n_chunks; c += 64) {
		const int i0 = c * SD_CH;
		double f0 = 0.0, f1 = 0.0;
		if (i0 < u.y_len) {
			// output i is the filter centred on sample i + 1: sum_q tap[q] y[i + 1 - hl + q] (hv_bandpass_kernel)
			const double *__restrict__ w = y + (i0 + 1 - hl);
			double prev = w[0];
			for (int q = 0; q < nt; ++q) {
				const double next = w[q + 1];
				f0 = fma(Tp[q], prev, f0);
				f1 = fma(Tp[q], next, f1);
				prev = next;
			}
		}
		out[2 * c] = f0;
		out[2 * c + 1] = f1;
	}
}

// Quiet chunks (round 5).  A sliding sum carries the rounding of everything that has passed through it: 1e-14 of the LOUDEST
// stretch since it was built.  Where the signal then falls silent -- digital silence behind an utterance, a gated segment; after
// decimation the decimator's decaying tail -- the true output drops by hundreds of decibels and the sum keeps oscillating at
// its own frequency with that stale amplitude: a perfectly periodic "signal" in every band, whose zero crossings make consistent
// raw candidates at the band frequencies and pull the contour's ends (9 Hz on the last frames of a voiced segment in front of a
// gated one, 24 / 96 kHz: profiles/r05_c_quiet_chunks.txt; the direct FIR sum and the reference's FFT convolution -- noise 1e-16 of
// the global maximum, incoherent -- leave nothing of the kind).  hv_blockmax_kernel / hv_quiet_kernel mark every chunk in
// whose span (build window included) the level of 64 samples falls below 1e-8 of what the span has seen before; the sliding
// kernels skip those, and hv_bandpass_quiet_kernel does them as direct FIR sums into the same slots (four times the work per
// chunk; none on signals with a noise floor, a chunk or two per pause on digitally silenced ones).
constexpr double kQuietDrop = 1e-8;
__global__ __launch_bounds__(256) void hv_blockmax_kernel(SdArgs a, double *__restrict__ bmax) {
	const int lane = threadIdx.x & 63, blk = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (blk >= a.n_blk) return;
	const HvUtt u = a.utts[blockIdx.y];
	const int i = blk * 64 + lane;
	double v = i < u.y_len ? fabs(a.y[u.y_off + i]) : 0.0;
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
	if (lane == 0) bmax[(long long)blockIdx.y * a.n_blk + blk] = v;
}
__global__ __launch_bounds__(64) void hv_quiet_kernel(SdArgs a, int *__restrict__ quiet, int enable) {
	const HvUtt u = a.utts[blockIdx.x];
	const double *__restrict__ bm = a.bmax + (long long)blockIdx.x * a.n_blk;
	int any = 0;
	for (int c = threadIdx.x; c < a.n_chunks; c += 64) {
		const int i0 = c * SD_CH;
		int flag = 0;
		if (enable && i0 < u.y_len) {
			// the samples a lane of this chunk ever sees: its build window (2 hl + 1 samples in front of output i0) to its lookahead
			// (not beyond the utterance's own end: what a lane reads there is margin, and its outputs there are masked)
			const int b0 = max(0, (i0 - 2 * a.hl_max - 2) / 64), b1 = min(min(a.n_blk - 1, (u.y_len - 1) / 64), (i0 + SD_CH + a.hl_max + 4) / 64);
			double m = 0.0;
			for (int b = b0; b <= b1; ++b) {
				const double v = bm[b];
				if (v < kQuietDrop * m) flag = 1;
				m = fmax(m, v);
			}
		}
		quiet[(long long)blockIdx.x * a.n_chunks + c] = flag;
		any |= flag;
	}
	any = __any(any);
	if (threadIdx.x == 0) quiet[(long long)gridDim.x * a.n_chunks + blockIdx.x] = any;
}
// the marked chunks of one (band, utterance): hv_bandpass_kernel's tile -- direct sums of the filter, eight outputs per thread,
// ordered compaction -- on whole chunks, edges into the chunk's slots, the chunk's first two outputs and its lookahead from
// hv_seam_kernel like everywhere else
__global__ __launch_bounds__(BP_T, 3) void hv_bandpass_quiet_kernel(SdArgs a) {
	__shared__ double Ys[(BP_TILE + 2 * HL_MAX + 32) * 9 / 8 + 16];
	__shared__ double Tp[2 * HL_MAX + 16];
	double *Ss = Ys;
	__shared__ unsigned long long scan_s[BP_T / 64];
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int band = blockIdx.x;
	if (!a.quiet[(long long)gridDim.y * a.n_chunks + blockIdx.y]) return;  // (nothing marked in this utterance: the usual case)
	const HvUtt u = a.utts[blockIdx.y];
	const int hl = a.half_len[band];
	const int nt8 = ((2 * hl + 1 + 7) / 8) * 8;
	const double *__restrict__ y = a.y + u.y_off;
	const double *__restrict__ taps = a.taps + a.tap_off[band];
	for (int i = tid; i < nt8; i += BP_T) Tp[i] = taps[i];
	const int cap = a.slot_cap[band];
	const int t0 = tid * BP_R;
	const int *__restrict__ qf = a.quiet + (long long)blockIdx.y * a.n_chunks;
	for (int c = 0; c < a.n_chunks; ++c) {
		if (!qf[c]) continue;  // (uniform)
		const int ts = c * SD_CH;
		double *__restrict__ slot = a.slots + blockIdx.y * a.slots_per_utt + a.slot_off[band] + (long long)c * 4 * cap;
		const double *__restrict__ sm = a.seam + (((long long)blockIdx.y * a.n_bands + band) * (a.n_chunks + 1) + c) * 2;
		// Ys[m] = y[ts + 1 - hl + m], m in [0, TILE + nt8 + 8)
		__syncthreads();
		for (int m = tid; m < BP_TILE + nt8 + 8; m += BP_T) {
			int g = ts + 1 - hl + m;
			Ys[padidx(m)] = (g >= 0 && g < u.y_len) ? y[g] : 0.0;
		}
		__syncthreads();
		double acc[BP_R];
#pragma unroll
		for (int j = 0; j < BP_R; ++j) acc[j] = 0.0;
		double w[16];
#pragma unroll
		for (int k = 0; k < 8; ++k) w[k] = Ys[padidx(t0 + k)];
		for (int q0 = 0; q0 < nt8; q0 += 8) {
			double tp[8];
#pragma unroll
			for (int k = 0; k < 8; ++k) { w[8 + k] = Ys[padidx(t0 + q0 + 8 + k)]; tp[k] = Tp[q0 + k]; }
#pragma unroll
			for (int qq = 0; qq < 8; ++qq)
#pragma unroll
				for (int j = 0; j < BP_R; ++j) acc[j] = fma(tp[qq], w[qq + j], acc[j]);
#pragma unroll
			for (int k = 0; k < 8; ++k) w[k] = w[8 + k];
		}
		__syncthreads();  // every thread is done reading the signal tile
#pragma unroll
		for (int j = 0; j < BP_R; ++j) Ss[padidx(t0 + j)] = acc[j];
		if (tid == 0) {  // one value per sample at the borders (hv_seam_kernel)
			Ss[padidx(0)] = sm[0];
			Ss[padidx(1)] = sm[1];
			Ss[padidx(BP_TILE)] = sm[2];
			Ss[padidx(BP_TILE + 1)] = sm[3];
		}
		__syncthreads();
		double sv[BP_R + 2];
#pragma unroll
		for (int j = 0; j < BP_R + 2; ++j) sv[j] = Ss[padidx(t0 + j)];
		unsigned int mask[4] = {0, 0, 0, 0};
#pragma unroll
		for (int j = 0; j < BP_R; ++j) {
			const int i = ts + t0 + j;
			const double s0 = sv[j], s1 = sv[j + 1], s2 = sv[j + 2];
			if (i + 1 < u.y_len) {
				if (0.0 < s0 && s1 <= 0.0) mask[0] |= 1u << j;
				if (0.0 < -s0 && -s1 <= 0.0) mask[1] |= 1u << j;
			}
			if (i + 2 < u.y_len) {
				const double d0 = s1 - s0, d1 = s2 - s1;
				if (0.0 < d0 && d1 <= 0.0) mask[2] |= 1u << j;
				if (0.0 < -d0 && -d1 <= 0.0) mask[3] |= 1u << j;
			}
		}
		unsigned long long packed = 0;
#pragma unroll
		for (int ty = 0; ty < 4; ++ty) packed |= (unsigned long long)__popc(mask[ty]) << (16 * ty);
		unsigned long long inc = packed;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			unsigned long long t = __shfl_up(inc, o, 64);
			if (lane >= o) inc += t;
		}
		if (lane == 63) scan_s[wv] = inc;
		__syncthreads();
		unsigned long long base = 0, total = 0;
#pragma unroll
		for (int k = 0; k < BP_T / 64; ++k) {
			unsigned long long t = scan_s[k];
			if (k < wv) base += t;
			total += t;
		}
		const unsigned long long excl = base + inc - packed;
#pragma unroll
		for (int ty = 0; ty < 4; ++ty) {
			int at = (int)((excl >> (16 * ty)) & 0xffffull);
#pragma unroll
			for (int j = 0; j < BP_R; ++j) {
				if (mask[ty] & (1u << j)) {
					const int i = ts + t0 + j;
					double v0, v1;
					if (ty < 2) { v0 = sv[j]; v1 = sv[j + 1]; }
					else { v0 = sv[j + 1] - sv[j]; v1 = sv[j + 2] - sv[j + 1]; }
					const double fine = (i + 1) - v0 / (v1 - v0);
					if (at < cap) slot[(long long)ty * cap + at] = fine;
					++at;
				}
			}
		}
		if (tid < 4) a.slot_count[(((long long)blockIdx.y * a.n_bands + band) * a.n_chunks + c) * 4 + tid] = (int)((total >> (16 * tid)) & 0xffffull);
	}
}

#ifndef WC_SDFT_U
#define WC_SDFT_U 4      // outputs per trip = depth of the sample prefetch
#endif
#ifndef WC_SDFT_WAVES
#define WC_SDFT_WAVES 1  // minimum wavefronts per SIMD asked of the register allocator
#endif
#ifndef WC_SDFT_RING
#define WC_SDFT_RING 1   // edges leave for their slots in whole 32-byte sectors (below)
#endif
// DEFER (round 6, default): the detectors' work is split.  Some lane of a wavefront has an edge at almost every sample, so the
// two blocks that turn an edge into its interpolated position and file it (quotient, counters, ring: ~34 instructions each) ran
// at almost every step with a lane or two active -- 68 of the 127 vector instructions of a step.  Now a step only marks its edges
// in a bit mask and stores its output (LDS, a row of 64 per step); after eight steps every lane walks its own marks: as many
// trips as the busiest lane has edges (one to five), each reading the three outputs around the edge back from LDS.  The same
// quotients of the same values filed in the same order: the same bits (test_bandpass_with_eight_lanes_per_band_is_bit_identical).
template <bool DEFER>
__global__ __launch_bounds__(64, WC_SDFT_WAVES) void hv_bandpass_sdft_kernel(SdArgs a) {
	const int lane = threadIdx.x;
	const HvUtt u = a.utts[blockIdx.y];
	const int item = blockIdx.x * 64 + lane;  // (chunk, band), band fastest: a wave holds neighbouring bands
	const bool valid = item < a.n_bands * a.n_chunks;
	const int chunk = valid ? item / a.n_bands : 0;
	const int band = valid ? item - chunk * a.n_bands : 0;
	const int i0 = chunk * SD_CH;
	const bool skip = valid && a.quiet[(long long)blockIdx.y * a.n_chunks + chunk] != 0;  // (left to hv_bandpass_quiet_kernel, counts and all)
	const bool live = valid && i0 < u.y_len && !skip;
	if (__ballot(live) == 0ull) {
		if (valid && !skip) {
			int *c = a.slot_count + (((long long)blockIdx.y * a.n_bands + band) * a.n_chunks + chunk) * 4;
			c[0] = c[1] = c[2] = c[3] = 0;
		}
		return;
	}
	const int hl = live ? a.half_len[band] : 0;
	const double *__restrict__ y = a.y + u.y_off;  // zero margins of Y_PADL / Y_PADR samples: no bounds checks below
	const int ylen = u.y_len;
	double2 R[7], D[7];
#pragma unroll
	for (int v = 0; v < 7; ++v) { R[v] = a.rot[band * 7 + v]; D[v] = make_double2(0.0, 0.0); }
	const double2 P = a.p0[band];
	// ---- first window: samples i0 + 1 - hl .. i0 + 1 + hl (m = i + 1 + hl is the newest sample of output i) ----
	int hlmax = hl;
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) hlmax = max(hlmax, __shfl_xor(hlmax, o, 64));
	const int qb = live ? i0 + 1 + hl - 2 * hlmax : 0;  // (>= -Y_PADL; idle lanes read their own margin)
	for (int sidx = 0; sidx <= 2 * hlmax; ++sidx) {
		const int q = qb + sidx;   // this lane's window starts at sidx = 2 (hlmax - hl)
		const double yv = (live && sidx >= 2 * (hlmax - hl)) ? y[q] : 0.0;
		const double ux = yv * P.x, uy = yv * P.y;
#pragma unroll
		for (int v = 0; v < 7; ++v) {
			const double nx = fma(R[v].x, D[v].x, fma(-R[v].y, D[v].y, ux));
			const double ny = fma(R[v].x, D[v].y, fma(R[v].y, D[v].x, uy));
			D[v] = make_double2(nx, ny);
		}
	}
	auto out = [&]() -> double {
		// Nuttall coefficients of reference src/world_common.cpp:118-126, halved for the +- pairs
		double f = 0.355768 * D[0].x;
		f = fma(-0.243698, D[1].x + D[2].x, f);
		f = fma(0.072116, D[3].x + D[4].x, f);
		f = fma(-0.006302, D[5].x + D[6].x, f);
		return f;
	};
	auto quot = [](double n, double d) -> double {  // n / d for |n| <= |d|, d a normal number
		double r = __builtin_amdgcn_rcp(d);
		r = fma(fma(-d, r, 1.0), r, r);
		r = fma(fma(-d, r, 1.0), r, r);
		const double q = n * r;
		return fma(fma(-d, q, n), r, q);
	};
	auto slide = [&](double yn, double yo) {  // from output i to output i + 1: yn = y[i + 2 + hl] enters, yo = y[i + 1 - hl] leaves
		const double ux = yn * P.x, uy = yn * P.y;
		const double vx = yo * P.x, vy = -(yo * P.y);
#pragma unroll
		for (int v = 0; v < 7; ++v) {
			const double tx = D[v].x - vx, ty = D[v].y - vy;
			const double nx = fma(R[v].x, tx, fma(-R[v].y, ty, ux));
			const double ny = fma(R[v].x, ty, fma(R[v].y, tx, uy));
			D[v] = make_double2(nx, ny);
		}
	};
	const double *__restrict__ pn = y + (live ? i0 + 2 + hl : 0), *__restrict__ po = y + (live ? i0 + 1 - hl : 0);
	double s0 = out();
	slide(pn[0], po[0]);
	double s1 = out();
	// the chunk's first two outputs and its two outputs of lookahead (the next chunk's first two) from hv_seam_kernel: see there
	const double *__restrict__ sm = a.seam + (((long long)blockIdx.y * a.n_bands + band) * (a.n_chunks + 1) + chunk) * 2;
	if (live) { s0 = sm[0]; s1 = sm[1]; }
	const double fn0 = live ? sm[2] : 0.0, fn1 = live ? sm[3] : 0.0;
	const int cap = a.slot_cap[band];
	double *__restrict__ slot = a.slots + blockIdx.y * a.slots_per_utt + a.slot_off[band] + (long long)chunk * 4 * cap;
	int cnt[4] = {0, 0, 0, 0};
	// An edge goes to its slot four at a time: a lone 8-byte store leaves a cache line that is evicted long before the lane's
	// next edge of that type arrives (a wave's lanes write 256 separate streams; 4.4 GB of write traffic per 64 utterances for
	// 1.1 GB of edges, one 32-byte sector per edge).  The lane parks edges c = 4 m .. 4 m + 2 in LDS and writes the whole
	// 32-byte sector when edge 4 m + 3 arrives; what is left at the end of the chunk follows then.  (cap is a multiple of 4 and
	// the slots are 32-byte aligned.)  WC_SDFT_RING=0: every edge stored as it appears.
#if WC_SDFT_RING
	__shared__ double ring[12 * 64];  // [type][c & 3 < 3][lane] (the fourth edge of a sector leaves with the three parked ones)
	auto put = [&](int ty, int c, double fine) {
		const int k = c & 3;
		if (k != 3) {
			ring[(ty * 3 + k) * 64 + lane] = fine;
		} else {
			const double e0 = ring[(ty * 3 + 0) * 64 + lane], e1 = ring[(ty * 3 + 1) * 64 + lane], e2 = ring[(ty * 3 + 2) * 64 + lane];
			double2 *dst = reinterpret_cast<double2 *>(slot + (long long)ty * cap + (c - 3));
			dst[0] = make_double2(e0, e1);
			dst[1] = make_double2(e2, fine);
		}
	};
#else
	auto put = [&](int ty, int c, double fine) { slot[(long long)ty * cap + c] = fine; };
#endif
	const int i_end = live ? min(i0 + SD_CH, ylen) : i0;
	int steps = i_end - i0;  // (lanes of one wave may sit in two different chunks)
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) steps = max(steps, __shfl_xor(steps, o, 64));
	// four outputs per trip; the samples of a trip are requested one trip ahead of their use
	constexpr int U = WC_SDFT_U;
	static_assert(U >= 2 && SD_CH % U == 0, "the last trip of a whole chunk ends on the chunk's border");
	double yn[U], yo[U];
#pragma unroll
	for (int k = 0; k < U; ++k) { yn[k] = pn[1 + k]; yo[k] = po[1 + k]; }
	if constexpr (DEFER) {
		static_assert(U == 4 && SD_CH % 8 == 0, "blocks of two trips");
		__shared__ double O[10 * 64];  // [output of the block, two in front of it][lane]
		// The signs of the newest output and difference are looked at once -- x <= 0 and x >= 0, two compares each, kept as lane masks in
		// scalar registers -- and serve the two steps that use them: "0 < s0" is (s0 >= 0) and not (s0 <= 0), so that a NaN (neither)
		// still marks nothing; the masks are combined by scalar instructions and come back as a per-lane select.
		typedef unsigned long long LaneMask;
		LaneMask le_b = __ballot(s0 <= 0.0), ge_b = __ballot(s0 >= 0.0);
		LaneMask p_a = ge_b & ~le_b, n_a = le_b & ~ge_b;                           // s0
		le_b = __ballot(s1 <= 0.0); ge_b = __ballot(s1 >= 0.0);                    // s1
		LaneMask p_d, n_d;                                                         // d0
		{
			const double dprev = s1 - s0;
			const LaneMask le = __ballot(dprev <= 0.0), ge = __ballot(dprev >= 0.0);
			p_d = ge & ~le; n_d = le & ~ge;
		}
		unsigned long long cnt4 = 0ull;  // the four counts, sixteen bits each (a chunk has 2048 samples)
		for (int st = 0; st < steps; st += 8) {
			O[lane] = s0;
			O[64 + lane] = s1;
			unsigned mask = 0u;  // bit k: a zero crossing at step k of the block, bit 8 + k: an extremum
#pragma unroll
			for (int half = 0; half < 2; ++half) {
				double cn[U], co[U];
#pragma unroll
				for (int k = 0; k < U; ++k) { cn[k] = yn[k]; co[k] = yo[k]; }
#pragma unroll
				for (int k = 0; k < U; ++k) { yn[k] = pn[st + 4 * half + U + 1 + k]; yo[k] = po[st + 4 * half + U + 1 + k]; }
				const bool last_trip = live && st + 4 * half + U == SD_CH;
#pragma unroll
				for (int k = 0; k < U; ++k) {
					const int kk = 4 * half + k;
					slide(cn[k], co[k]);
					double s2 = out();
					if (k == U - 2) s2 = last_trip ? fn0 : s2;
					if (k == U - 1) s2 = last_trip ? fn1 : s2;
					O[(kk + 2) * 64 + lane] = s2;
					const double d1 = s2 - s1;
					const LaneMask le_e = __ballot(d1 <= 0.0), ge_e = __ballot(d1 >= 0.0);
					const bool zc = __builtin_amdgcn_inverse_ballot_w64((p_a & le_b) | (n_a & ge_b));
					const bool ex = __builtin_amdgcn_inverse_ballot_w64((p_d & le_e) | (n_d & ge_e));
					mask |= (zc ? 1u << kk : 0u) | (ex ? 0x100u << kk : 0u);
					p_a = ge_b & ~le_b; n_a = le_b & ~ge_b;
					le_b = __ballot(s2 <= 0.0); ge_b = __ballot(s2 >= 0.0);
					p_d = ge_e & ~le_e; n_d = le_e & ~ge_e;
					s0 = s1;
					s1 = s2;
				}
			}
			{
				// the steps of the block that lie inside the chunk and have their one / two outputs of lookahead inside the signal
				const int ib = i0 + st;
				const int n1 = min(8, max(0, min(i_end, ylen - 1) - ib)), n2 = min(8, max(0, min(i_end, ylen - 2) - ib));
				mask &= ((1u << n1) - 1u) | (((1u << n2) - 1u) << 8);
			}
			rq_fence();
			while (__ballot(mask != 0u) != 0ull) {
				if (mask != 0u) {
					const int b = __ffs((int)mask) - 1;
					mask &= mask - 1u;
					const bool is_ex = b >= 8;
					const int kk = b & 7;
					const double a0 = O[kk * 64 + lane], a1 = O[(kk + 1) * 64 + lane], a2 = O[(kk + 2) * 64 + lane];
					const double d0 = a1 - a0, d1 = a2 - a1;
					const double num = is_ex ? d0 : a0, den = is_ex ? d1 - d0 : d0;
					const double fine = (i0 + st + kk + 1) - quot(num, den);
					const int ty = (is_ex ? 2 : 0) + (0.0 < num ? 0 : 1);  // negative-going, positive-going, peak, dip
					const int c = (int)(cnt4 >> (16 * ty)) & 0xffff;
					if (c < cap) put(ty, c, fine);
					cnt4 += 1ull << (16 * ty);
				}
			}
			rq_fence();
		}
#pragma unroll
		for (int ty = 0; ty < 4; ++ty) cnt[ty] = (int)(cnt4 >> (16 * ty)) & 0xffff;
	} else
	for (int st = 0; st < steps; st += U) {
		double cn[U], co[U];
#pragma unroll
		for (int k = 0; k < U; ++k) { cn[k] = yn[k]; co[k] = yo[k]; }
#pragma unroll
		for (int k = 0; k < U; ++k) { yn[k] = pn[st + U + 1 + k]; yo[k] = po[st + U + 1 + k]; }
		const bool last_trip = live && st + U == SD_CH;  // (a whole chunk's last trip: its outputs 2 and 3 are the next chunk's first two)
#pragma unroll
		for (int k = 0; k < U; ++k) {
			const int i = i0 + st + k;
			slide(cn[k], co[k]);
			double s2 = out();
			if (k == U - 2) s2 = last_trip ? fn0 : s2;
			if (k == U - 1) s2 = last_trip ? fn1 : s2;
			// zeroCrossingEngine (reference :1179-1219) over the y_length samples (types 0, 1) and over the
			// y_length - 1 first differences (types 2, 3); fine edge = edges[e] - sig[e-1] / (sig[e] - sig[e-1]) with
			// edges = i + 1, the same for a signal and its negation
			const double d0 = s1 - s0, d1 = s2 - s1;
			const bool in1 = i < i_end && i + 1 < ylen, in2 = i < i_end && i + 2 < ylen;
			const bool neg = in1 && 0.0 < s0 && s1 <= 0.0, pos = in1 && 0.0 < -s0 && -s1 <= 0.0;
			const bool pk = in2 && 0.0 < d0 && d1 <= 0.0, dp = in2 && 0.0 < -d0 && -d1 <= 0.0;
			// (some lane of the wave has an edge at almost every step, so these blocks run all the time: selects between
			// two counters instead of an indexed array, and a short reciprocal-based quotient -- the ratio lies in [-1, 0]
			// and is good to an ulp, far below the band-pass's own rounding)
			if (neg || pos) {
				const double fine = (i + 1) - quot(s0, d0);
				const int c = neg ? cnt[0] : cnt[1];
				if (c < cap) put(neg ? 0 : 1, c, fine);
				cnt[0] += neg ? 1 : 0;
				cnt[1] += pos ? 1 : 0;
			}
			if (pk || dp) {
				const double fine = (i + 1) - quot(d0, d1 - d0);
				const int c = pk ? cnt[2] : cnt[3];
				if (c < cap) put(pk ? 2 : 3, c, fine);
				cnt[2] += pk ? 1 : 0;
				cnt[3] += dp ? 1 : 0;
			}
			s0 = s1;
			s1 = s2;
		}
	}
#if WC_SDFT_RING
	if (live) {
#pragma unroll
		for (int ty = 0; ty < 4; ++ty) {
			const int stored = min(cnt[ty], cap), rem = stored & 3;
			for (int k = 0; k < rem; ++k) slot[(long long)ty * cap + (stored - rem) + k] = ring[(ty * 3 + k) * 64 + lane];
		}
	}
#endif
	if (valid && !skip) {
		int *c = a.slot_count + (((long long)blockIdx.y * a.n_bands + band) * a.n_chunks + chunk) * 4;
#pragma unroll
		for (int ty = 0; ty < 4; ++ty) c[ty] = cnt[ty];
	}
}

// The same arithmetic with the seven sliding sums of a (band, chunk) in seven lanes of a group of eight (small batches: a
// single utterance has only 95 wavefronts of the kernel above, each a serial walk over ~2800 samples -- 0.76 ms with most of
// the chip idle).  Lane s of a group holds D_v for v = 0, -, 1, 2, 3, 4, 5, 6: every recurrence runs the instructions it runs
// above, the +- pairs of the Nuttall sum meet through one quad exchange (an addition is commutative), lane 0 collects the three
// pair sums in the order of out() above and runs the detectors.  Same bits in the slots, eight times the wavefronts, about
// half the instructions per sample and wavefront.
template <int X>
__device__ __forceinline__ double sd8_xor(double v) {
	int w[2] = {__double2loint(v), __double2hiint(v)};
#pragma unroll
	for (int k = 0; k < 2; ++k) {
		if (X == 1) w[k] = __builtin_amdgcn_mov_dpp(w[k], 0xB1, 0xF, 0xF, true);       // quad_perm:[1,0,3,2]
		else if (X == 2) w[k] = __builtin_amdgcn_mov_dpp(w[k], 0x4E, 0xF, 0xF, true);  // quad_perm:[2,3,0,1]
		else {
			int r = __builtin_amdgcn_update_dpp(w[k], w[k], 0x104, 0xF, 0x5, false);   // row_shl:4 -> lanes 0-3, 8-11
			w[k] = __builtin_amdgcn_update_dpp(r, w[k], 0x114, 0xF, 0xA, false);       // row_shr:4 -> lanes 4-7, 12-15
		}
	}
	return __hiloint2double(w[1], w[0]);
}

__global__ __launch_bounds__(64) void hv_bandpass_sdft8_kernel(SdArgs a) {
	const int lane = threadIdx.x, sub = lane & 7;
	const HvUtt u = a.utts[blockIdx.y];
	const int item = blockIdx.x * 8 + (lane >> 3);
	const bool valid = item < a.n_bands * a.n_chunks;
	const int chunk = valid ? item / a.n_bands : 0;
	const int band = valid ? item - chunk * a.n_bands : 0;
	const int i0 = chunk * SD_CH;
	const bool skip = valid && a.quiet[(long long)blockIdx.y * a.n_chunks + chunk] != 0;
	const bool live = valid && i0 < u.y_len && !skip;
	const bool head = sub == 0;
	if (__ballot(live) == 0ull) {
		if (valid && head && !skip) {
			int *c = a.slot_count + (((long long)blockIdx.y * a.n_bands + band) * a.n_chunks + chunk) * 4;
			c[0] = c[1] = c[2] = c[3] = 0;
		}
		return;
	}
	const int hl = live ? a.half_len[band] : 0;
	const double *__restrict__ y = a.y + u.y_off;
	const int ylen = u.y_len;
	const int v = sub == 0 ? 0 : sub - 1;  // (lane 1 idles on a copy of v = 0; its sum is never read)
	const double2 R = a.rot[band * 7 + v];
	double2 D = make_double2(0.0, 0.0);
	const double2 P = a.p0[band];
	int hlmax = hl;
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) hlmax = max(hlmax, __shfl_xor(hlmax, o, 64));
	const int qb = live ? i0 + 1 + hl - 2 * hlmax : 0;
	for (int sidx = 0; sidx <= 2 * hlmax; ++sidx) {
		const int q = qb + sidx;
		const double yv = (live && sidx >= 2 * (hlmax - hl)) ? y[q] : 0.0;
		const double ux = yv * P.x, uy = yv * P.y;
		const double nx = fma(R.x, D.x, fma(-R.y, D.y, ux));
		const double ny = fma(R.x, D.y, fma(R.y, D.x, uy));
		D = make_double2(nx, ny);
	}
	auto out_of = [&](double dx) -> double {  // (meaningful in lane 0 of the group)
		const double pr = head ? dx : dx + sd8_xor<1>(dx);  // lanes 2, 4, 6: the +- pairs
		const double p2 = sd8_xor<2>(pr), p4 = sd8_xor<4>(pr), p6 = sd8_xor<4>(p2);
		double f = 0.355768 * pr;
		f = fma(-0.243698, p2, f);
		f = fma(0.072116, p4, f);
		f = fma(-0.006302, p6, f);
		return f;
	};
	auto out = [&]() -> double { return out_of(D.x); };
	auto quot = [](double n, double d) -> double {
		double r = __builtin_amdgcn_rcp(d);
		r = fma(fma(-d, r, 1.0), r, r);
		r = fma(fma(-d, r, 1.0), r, r);
		const double q = n * r;
		return fma(fma(-d, q, n), r, q);
	};
	auto slide = [&](double yn, double yo) {
		const double ux = yn * P.x, uy = yn * P.y;
		const double vx = yo * P.x, vy = -(yo * P.y);
		const double tx = D.x - vx, ty = D.y - vy;
		const double nx = fma(R.x, tx, fma(-R.y, ty, ux));
		const double ny = fma(R.x, ty, fma(R.y, tx, uy));
		D = make_double2(nx, ny);
	};
	const double *__restrict__ pn = y + (live ? i0 + 2 + hl : 0), *__restrict__ po = y + (live ? i0 + 1 - hl : 0);
	double s0 = out();
	slide(pn[0], po[0]);
	double s1 = out();
	// the chunk's first two outputs and its two outputs of lookahead (the next chunk's first two) from hv_seam_kernel: see there
	const double *__restrict__ sm = a.seam + (((long long)blockIdx.y * a.n_bands + band) * (a.n_chunks + 1) + chunk) * 2;
	if (live) { s0 = sm[0]; s1 = sm[1]; }
	const double fn0 = live ? sm[2] : 0.0, fn1 = live ? sm[3] : 0.0;
	const int cap = a.slot_cap[band];
	double *__restrict__ slot = a.slots + blockIdx.y * a.slots_per_utt + a.slot_off[band] + (long long)chunk * 4 * cap;
	int cnt[4] = {0, 0, 0, 0};
	const int i_end = live ? min(i0 + SD_CH, ylen) : i0;
	int steps = i_end - i0;
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) steps = max(steps, __shfl_xor(steps, o, 64));
	constexpr int U = WC_SDFT_U;
	static_assert(U >= 2 && SD_CH % U == 0, "the last trip of a whole chunk ends on the chunk's border");
	double yn[U], yo[U];
#pragma unroll
	for (int k = 0; k < U; ++k) { yn[k] = pn[1 + k]; yo[k] = po[1 + k]; }
	for (int st = 0; st < steps; st += U) {
		double cn[U], co[U];
#pragma unroll
		for (int k = 0; k < U; ++k) { cn[k] = yn[k]; co[k] = yo[k]; }
#pragma unroll
		for (int k = 0; k < U; ++k) { yn[k] = pn[st + U + 1 + k]; yo[k] = po[st + U + 1 + k]; }
		// the sums of the trip first (a chain of dependent operations per sum), then the outputs, then the detectors: the
		// branches of the detectors would otherwise fence every step's chain off from the next one's
		double sx[U];
#pragma unroll
		for (int k = 0; k < U; ++k) {
			slide(cn[k], co[k]);
			sx[k] = D.x;
		}
		double so[U];
#pragma unroll
		for (int k = 0; k < U; ++k) so[k] = out_of(sx[k]);
		if (live && st + U == SD_CH) { so[U - 2] = fn0; so[U - 1] = fn1; }  // (the next chunk's first two outputs: hv_seam_kernel)
#pragma unroll
		for (int k = 0; k < U; ++k) {
			const int i = i0 + st + k;
			const double s2 = so[k];
			const double d0 = s1 - s0, d1 = s2 - s1;
			const bool in1 = head && i < i_end && i + 1 < ylen, in2 = head && i < i_end && i + 2 < ylen;
			const bool neg = in1 && 0.0 < s0 && s1 <= 0.0, pos = in1 && 0.0 < -s0 && -s1 <= 0.0;
			const bool pk = in2 && 0.0 < d0 && d1 <= 0.0, dp = in2 && 0.0 < -d0 && -d1 <= 0.0;
			if (neg || pos) {
				const double fine = (i + 1) - quot(s0, d0);
				const int c = neg ? cnt[0] : cnt[1];
				if (c < cap) slot[(neg ? 0 : cap) + c] = fine;
				cnt[0] += neg ? 1 : 0;
				cnt[1] += pos ? 1 : 0;
			}
			if (pk || dp) {
				const double fine = (i + 1) - quot(d0, d1 - d0);
				const int c = pk ? cnt[2] : cnt[3];
				if (c < cap) slot[(pk ? 2 * cap : 3 * cap) + c] = fine;
				cnt[2] += pk ? 1 : 0;
				cnt[3] += dp ? 1 : 0;
			}
			s0 = s1;
			s1 = s2;
		}
	}
	if (valid && head && !skip) {
		int *c = a.slot_count + (((long long)blockIdx.y * a.n_bands + band) * a.n_chunks + chunk) * 4;
#pragma unroll
		for (int ty = 0; ty < 4; ++ty) c[ty] = cnt[ty];
	}
}

// The eight-lane kernel in blocks of 64 samples (round 6).  Above, every step of a group runs the recurrence, then gathers the seven
// sums through quad exchanges, then runs the detectors in lane 0 with the other seven lanes looking on: ~50 instructions per
// sample, every one on the critical path of a wavefront that has its SIMD to itself (one utterance: 0.55 ms of 2.1).  Here a
// step is the recurrence and one LDS store of the lane's sum; after 64 steps the block's outputs and detectors are done ACROSS
// the lanes -- lane r of a group takes samples 8 r .. 8 r + 7: the Nuttall sum in out()'s order from the seven rows, the two
// preceding outputs from lane r - 1 (lane 0: the previous block's last two, or the seam's), the detectors as above; the edges go
// to their slots in time order through a scan of the lanes' counts.  ~12 instructions per sample and wavefront in the steps,
// ~5 in the blocks; the same values in the same slots.  Rows of 64 sums, one pad per eight, stride 73: the stores of a step and
// the loads of a block are conflict-free.
constexpr int SD8_ROW = 73;
__global__ __launch_bounds__(64) void hv_bandpass_sdft8b_kernel(SdArgs a) {
	__shared__ double X[64 * SD8_ROW];
	const int lane = threadIdx.x, sub = lane & 7, grp = lane >> 3;
	const HvUtt u = a.utts[blockIdx.y];
	const int item = blockIdx.x * 8 + grp;
	const bool valid = item < a.n_bands * a.n_chunks;
	const int chunk = valid ? item / a.n_bands : 0;
	const int band = valid ? item - chunk * a.n_bands : 0;
	const int i0 = chunk * SD_CH;
	const bool skip = valid && a.quiet[(long long)blockIdx.y * a.n_chunks + chunk] != 0;
	const bool live = valid && i0 < u.y_len && !skip;
	const bool head = sub == 0;
	if (__ballot(live) == 0ull) {
		if (valid && head && !skip) {
			int *c = a.slot_count + (((long long)blockIdx.y * a.n_bands + band) * a.n_chunks + chunk) * 4;
			c[0] = c[1] = c[2] = c[3] = 0;
		}
		return;
	}
	const int hl = live ? a.half_len[band] : 0;
	const double *__restrict__ y = a.y + u.y_off;
	const int ylen = u.y_len;
	const int v = sub == 0 ? 0 : sub - 1;  // (lane 1 idles on a copy of v = 0; its row is never read)
	const double2 R = a.rot[band * 7 + v];
	double2 D = make_double2(0.0, 0.0);
	const double2 P = a.p0[band];
	int hlmax = hl;
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) hlmax = max(hlmax, __shfl_xor(hlmax, o, 64));
	const int qb = live ? i0 + 1 + hl - 2 * hlmax : 0;
	for (int sidx = 0; sidx <= 2 * hlmax; ++sidx) {
		const int q = qb + sidx;
		const double yv = (live && sidx >= 2 * (hlmax - hl)) ? y[q] : 0.0;
		const double ux = yv * P.x, uy = yv * P.y;
		const double nx = fma(R.x, D.x, fma(-R.y, D.y, ux));
		const double ny = fma(R.x, D.y, fma(R.y, D.x, uy));
		D = make_double2(nx, ny);
	}
	auto quot = [](double n, double d) -> double {
		double r = __builtin_amdgcn_rcp(d);
		r = fma(fma(-d, r, 1.0), r, r);
		r = fma(fma(-d, r, 1.0), r, r);
		const double q = n * r;
		return fma(fma(-d, q, n), r, q);
	};
	auto slide = [&](double yn, double yo) {
		const double ux = yn * P.x, uy = yn * P.y;
		const double vx = yo * P.x, vy = -(yo * P.y);
		const double tx = D.x - vx, ty = D.y - vy;
		const double nx = fma(R.x, tx, fma(-R.y, ty, ux));
		const double ny = fma(R.x, ty, fma(R.y, tx, uy));
		D = make_double2(nx, ny);
	};
	const double *__restrict__ pn = y + (live ? i0 + 2 + hl : 0), *__restrict__ po = y + (live ? i0 + 1 - hl : 0);
	slide(pn[0], po[0]);
	// the chunk's first two outputs and its two outputs of lookahead (the next chunk's first two) from hv_seam_kernel: see there
	const double *__restrict__ sm = a.seam + (((long long)blockIdx.y * a.n_bands + band) * (a.n_chunks + 1) + chunk) * 2;
	double c0 = live ? sm[0] : 0.0, c1 = live ? sm[1] : 0.0;  // the two outputs in front of the block
	const double fn0 = live ? sm[2] : 0.0, fn1 = live ? sm[3] : 0.0;
	const int cap = a.slot_cap[band];
	double *__restrict__ slot = a.slots + blockIdx.y * a.slots_per_utt + a.slot_off[band] + (long long)chunk * 4 * cap;
	int cnt[4] = {0, 0, 0, 0};
	const int i_end = live ? min(i0 + SD_CH, ylen) : i0;
	int steps = i_end - i0;
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) steps = max(steps, __shfl_xor(steps, o, 64));
	constexpr int U = 8;
	double *__restrict__ xw = X + lane * SD8_ROW;              // this lane's row, written a step at a time
	const double *__restrict__ xr = X + grp * 8 * SD8_ROW + 9 * sub;  // the group's rows at this lane's eight samples
	double yn[U], yo[U];
#pragma unroll
	for (int k = 0; k < U; ++k) { yn[k] = pn[1 + k]; yo[k] = po[1 + k]; }
	for (int st = 0; st < steps; st += 64) {
#pragma unroll
		for (int t = 0; t < 64 / U; ++t) {
			double cn[U], co[U];
#pragma unroll
			for (int k = 0; k < U; ++k) { cn[k] = yn[k]; co[k] = yo[k]; }
#pragma unroll
			for (int k = 0; k < U; ++k) { yn[k] = pn[st + t * U + U + 1 + k]; yo[k] = po[st + t * U + U + 1 + k]; }
#pragma unroll
			for (int k = 0; k < U; ++k) {
				slide(cn[k], co[k]);
				xw[t * 9 + k] = D.x;
			}
		}
		rq_fence();
		double o[10];
#pragma unroll
		for (int jj = 0; jj < 8; ++jj) {
			const double d0 = xr[jj];
			const double p2 = xr[2 * SD8_ROW + jj] + xr[3 * SD8_ROW + jj];
			const double p4 = xr[4 * SD8_ROW + jj] + xr[5 * SD8_ROW + jj];
			const double p6 = xr[6 * SD8_ROW + jj] + xr[7 * SD8_ROW + jj];
			double f = 0.355768 * d0;
			f = fma(-0.243698, p2, f);
			f = fma(0.072116, p4, f);
			f = fma(-0.006302, p6, f);
			o[2 + jj] = f;
		}
		rq_fence();
		if (live && st + 64 == SD_CH && sub == 7) { o[8] = fn0; o[9] = fn1; }  // (the next chunk's first two outputs: hv_seam_kernel)
		{
			const double b0 = __shfl_up(o[8], 1, 64), b1 = __shfl_up(o[9], 1, 64);
			o[0] = head ? c0 : b0;
			o[1] = head ? c1 : b1;
			c0 = __shfl(o[8], lane | 7, 64);
			c1 = __shfl(o[9], lane | 7, 64);
		}
		unsigned m_neg = 0, m_pos = 0, m_pk = 0, m_dp = 0;
		const int ib = i0 + st + 8 * sub;
#pragma unroll
		for (int jj = 0; jj < 8; ++jj) {
			const int i = ib + jj;
			const double s0 = o[jj], s1 = o[jj + 1], s2 = o[jj + 2];
			const double d0 = s1 - s0, d1 = s2 - s1;
			const bool in1 = i < i_end && i + 1 < ylen, in2 = i < i_end && i + 2 < ylen;
			const bool neg = in1 && 0.0 < s0 && s1 <= 0.0, pos = in1 && 0.0 < -s0 && -s1 <= 0.0;
			const bool pk = in2 && 0.0 < d0 && d1 <= 0.0, dp = in2 && 0.0 < -d0 && -d1 <= 0.0;
			m_neg |= (neg ? 1u : 0u) << jj;
			m_pos |= (pos ? 1u : 0u) << jj;
			m_pk |= (pk ? 1u : 0u) << jj;
			m_dp |= (dp ? 1u : 0u) << jj;
		}
		const unsigned mine = __popc(m_neg) | (__popc(m_pos) << 8) | (__popc(m_pk) << 16) | (__popc(m_dp) << 24);
		if (__ballot(mine != 0u) != 0ull) {
			unsigned inc = mine;  // counts of 0..8 per byte: a group's sums stay below 256
#pragma unroll
			for (int d = 1; d < 8; d <<= 1) {
				const unsigned t = __shfl_up(inc, d, 8);
				if (sub >= d) inc += t;
			}
			const unsigned before = inc - mine, total = __shfl(inc, 7, 8);
			int at[4];
#pragma unroll
			for (int ty = 0; ty < 4; ++ty) at[ty] = cnt[ty] + (int)((before >> (8 * ty)) & 255u);
			if (mine != 0u) {
#pragma unroll
				for (int jj = 0; jj < 8; ++jj) {
					const int i = ib + jj;
					const double s0 = o[jj], s1 = o[jj + 1], s2 = o[jj + 2];
					const double d0 = s1 - s0, d1 = s2 - s1;
					const bool neg = (m_neg >> jj) & 1u, pos = (m_pos >> jj) & 1u, pk = (m_pk >> jj) & 1u, dp = (m_dp >> jj) & 1u;
					if (neg || pos) {
						const double fine = (i + 1) - quot(s0, d0);
						const int c = neg ? at[0] : at[1];
						if (c < cap) slot[(neg ? 0 : cap) + c] = fine;
						at[0] += neg ? 1 : 0;
						at[1] += pos ? 1 : 0;
					}
					if (pk || dp) {
						const double fine = (i + 1) - quot(d0, d1 - d0);
						const int c = pk ? at[2] : at[3];
						if (c < cap) slot[(pk ? 2 * cap : 3 * cap) + c] = fine;
						at[2] += pk ? 1 : 0;
						at[3] += dp ? 1 : 0;
					}
				}
			}
#pragma unroll
			for (int ty = 0; ty < 4; ++ty) cnt[ty] += (int)((total >> (8 * ty)) & 255u);
		}
	}
	if (valid && head && !skip) {
		int *c = a.slot_count + (((long long)blockIdx.y * a.n_bands + band) * a.n_chunks + chunk) * 4;
#pragma unroll
		for (int ty = 0; ty < 4; ++ty) c[ty] = cnt[ty];
	}
}

struct CpArgs {
	const HvUtt *utts;
	const long long *slot_off;
	const int *slot_cap;
	long long slots_per_utt;
	const double *slots;
	const int *slot_count;
	const long long *ev_band_off;
	const int *ev_cap;
	double *events;
	int *ev_count;
	int *overflow;
	int *tile_run;  // [utt][band][n_chunks + 1][4]
	int n_bands, n_chunks;
	int copy;       // 0: the running counts only -- hv_raw reads the edges out of the slots itself
};

// One workgroup per (band, utterance), wave ty packs the slots of type ty chunk after chunk.
__global__ __launch_bounds__(256) void hv_compact_kernel(CpArgs a) {
	const int lane = threadIdx.x & 63, ty = threadIdx.x >> 6;
	const int band = blockIdx.x;
	const HvUtt u = a.utts[blockIdx.y];
	const int scap = a.slot_cap[band], cap = a.ev_cap[band];
	const double *__restrict__ slot = a.slots + blockIdx.y * a.slots_per_utt + a.slot_off[band];
	const int *__restrict__ sc = a.slot_count + ((long long)blockIdx.y * a.n_bands + band) * a.n_chunks * 4;
	double *__restrict__ ev = a.events + u.ev_off + a.ev_band_off[band] + (long long)ty * cap;
	int *__restrict__ trun = a.tile_run + ((long long)blockIdx.y * a.n_bands + band) * (a.n_chunks + 1) * 4;
	int run = 0;
	bool ovf = false;
	for (int c = 0; c < a.n_chunks; ++c) {
		if (lane == 0) trun[c * 4 + ty] = run;
		const int n = sc[c * 4 + ty];
		ovf = ovf || n > scap;
		const int m = min(n, scap);
		if (a.copy) {
			const double *__restrict__ src = slot + ((long long)c * 4 + ty) * scap;
			for (int j = lane; j < m; j += 64)
				if (run + j < cap) ev[run + j] = src[j];
		}
		run += n;
	}
	if (lane == 0) {
		trun[a.n_chunks * 4 + ty] = run;
		a.ev_count[((long long)blockIdx.y * a.n_bands + band) * 4 + ty] = run;
		if (ovf || (a.copy && run > cap)) atomicExch(a.overflow, 1);
	}
}

// ------------------------------------------------------------------------------------------------
// raw candidates on the 1 ms grid
// ------------------------------------------------------------------------------------------------
struct RawArgs {
	const HvUtt *utts;
	const double *events;
	const long long *ev_band_off;
	const int *ev_cap;
	const int *ev_count;
	const double *band_f0;
	const int *tile_run;
	int n_tiles, tile_adv;  // tile_run[q] = edges before sample q * tile_adv
	// SLOTS: the edges still lie in the (band, chunk, type) slots the sliding band-pass wrote them to; edge q of a list is entry
	// q - tile_run[c] of the chunk c with tile_run[c] <= q < tile_run[c + 1]
	const double *slots;
	const long long *slot_off;
	const int *slot_cap;
	long long slots_per_utt;
	double *raw;  // [utt: l1_off * n_bands][band][L1]
	int n_bands;
	double fs_d, f0_floor, f0_ceil;
	double r_fs_d;  // 1 / fs_d (div_const)
	const int4 *desc = nullptr;  // [utt][band][block of frames][type][2]: {base, end, first chunk, mode}, {running counts of that chunk and the three behind it} (hv_rawdesc_kernel)
	int desc_blocks = 0;
};

// c = #{k < n : loc[k] <= t} for the interval midpoints loc[k] = (e[k] + e[k+1]) / 2 / fs, searched in [lo, hi)
template <class E>
__device__ __forceinline__ int hv_count_le(E e, int lo, int hi, double fs, double t) {
	// The reference's test is (e[k] + e[k+1]) / 2.0 / fs <= t, two divisions per probe.  Bisect with the
	// division-free equivalent e[k] + e[k+1] <= 2 fs t (it can differ only when the two sides are within
	// rounding of each other), then settle the boundary with the exact test.
	const int lo0 = lo, hi0 = hi;
	const double tt = 2.0 * fs * t;
	while (lo < hi) {
		const int mid = (lo + hi) >> 1;
		if (e(mid) + e(mid + 1) <= tt) lo = mid + 1; else hi = mid;
	}
	auto exact = [&](int k) { return (e(k) + e(k + 1)) / 2.0 / fs <= t; };
	while (lo < hi0 && exact(lo)) ++lo;
	while (lo > lo0 && !exact(lo - 1)) --lo;
	return lo;
}

constexpr int RAW_T = 256;      // frames per workgroup
#ifndef WC_RAW_LDS
#define WC_RAW_LDS 304
#endif
constexpr int RAW_LDS = WC_RAW_LDS;    // staged intervals per type: 256 ms of a 968 Hz band hold 248 (+ 8 of margin); 19.5 KB per workgroup, 7 per CU (512: 4 per CU)

// One workgroup per (utterance, band, 256 consecutive 1 ms frames).  The fine edges that can matter for these
// frames form a short contiguous slice of each of the four event lists; wave `ty` locates the slice of type `ty`
// and stages its intervals in LDS as (midpoint time, interval frequency) pairs -- the divisions of reference
// :1210-1213 done once per interval instead of once per frame and probe -- then every thread interpolates its
// frame from LDS (falling back to the global lists if a slice does not fit).
// SLOTS (default): the lists are never materialised.  Edge q of a list is entry q - tile_run[c] of the slot of the chunk c with
// tile_run[c] <= q < tile_run[c + 1]; a block of 256 frames is 256 ms -- one chunk of the 8 kHz signal -- so its slice is its own
// chunk's slot plus four edges from either side, and both edges of an interval are fetched straight from their slots (a slice that
// reaches further than four chunks -- silence -- goes through a copy in LDS; single edges for the fallback are found by bisecting
// tile_run).  hv_compact_kernel then only forms the running counts: 0.33 -> 0.02 ms per half batch and 1.6 GB less traffic per
// step, for 0.13 ms more in this kernel.  <false>: per-band lists, as the FIR band-pass writes them and WC_HARVEST_RAW=lists packs them.
// How many staged intervals lie at or before frame i of the block (X[j] <= i / 1000.0)?  Every frame used to bisect X for it, eight
// dependent look-ups per type.  Round 6: interval j marks the FIRST frame at or behind its midpoint with j + 1 and a running
// maximum over the block's 256 frames gives every frame its count -- the same comparison, made once per interval.  One wavefront.
__device__ __forceinline__ void raw_frame_counts(const double *__restrict__ Xs, unsigned short *__restrict__ lo_, int len, int i0, int lane) {
	*reinterpret_cast<uint2 *>(&lo_[4 * lane]) = make_uint2(0u, 0u);
	rq_fence();
	for (int j0 = 0; j0 < len; j0 += 64) {
		const int j = j0 + lane;
		int rel = RAW_T;
		if (j < len) {
			const double x = Xs[j];
			int f = (int)ceil(x * 1000.0);
			while (f > 0 && div_const((double)(f - 1), 1000.0, 1.0 / 1000.0) >= x) --f;   // the first frame with f / 1000.0 >= x, exactly
			while (div_const((double)f, 1000.0, 1.0 / 1000.0) < x) ++f;
			rel = max(f - i0, 0);
		}
		// (two midpoints less than a millisecond apart may mark one frame: the later interval must win, whatever the order the
		// hardware serves the lanes of a store in)
		bool pending = rel < RAW_T;
		while (__ballot(pending) != 0ull) {
			if (pending) lo_[rel] = (unsigned short)(j + 1);
			rq_fence();
			pending = pending && lo_[rel] < j + 1;
		}
	}
	rq_fence();
	const uint2 w2 = *reinterpret_cast<const uint2 *>(&lo_[4 * lane]);
	int m0 = (int)(w2.x & 0xFFFFu), m1 = (int)(w2.x >> 16), m2 = (int)(w2.y & 0xFFFFu), m3 = (int)(w2.y >> 16);
	m1 = max(m1, m0); m2 = max(m2, m1); m3 = max(m3, m2);
	int inc = m3;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const int t = __shfl_up(inc, o, 64);
		if (lane >= o) inc = max(inc, t);
	}
	int ex = __shfl_up(inc, 1, 64);
	if (lane == 0) ex = 0;
	m0 = max(m0, ex); m1 = max(m1, ex); m2 = max(m2, ex); m3 = max(m3, ex);
	*reinterpret_cast<uint2 *>(&lo_[4 * lane]) = make_uint2((unsigned)m0 | ((unsigned)m1 << 16), (unsigned)m2 | ((unsigned)m3 << 16));
}

// The slice of edges a block of 256 frames needs of each of the four lists (see hv_raw_kernel), worked out once per (utterance,
// band, block, type) by a thread of its own (round 6): {first edge, one past the last, the chunk that holds the first, mode} and the
// running counts of that chunk and the three behind it.  mode 0: the slice lies within those four chunks (the usual case);
// 1: too long to stage; 2: it reaches further (silence) -- the block goes through the running counts itself.
__global__ __launch_bounds__(256) void hv_rawdesc_kernel(RawArgs a, int4 *__restrict__ desc, int nblk, int n_utt) {
	const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
	const long long total = 4ll * nblk * a.n_bands * n_utt;
	if (gid >= total) return;
	const int ty = (int)(gid & 3);
	const int b = (int)((gid >> 2) % nblk);
	const long long ub = (gid >> 2) / nblk;  // utterance * n_bands + band
	const HvUtt u = a.utts[ub / a.n_bands];
	int4 d0 = make_int4(0, 0, 0, 1), d1 = make_int4(0, 0, 0, 0);
	const int i0 = b * RAW_T;
	if (i0 < u.L1) {
		const int *__restrict__ trun = a.tile_run + ub * (a.n_tiles + 1) * 4;
		const int cnt_ty = a.ev_count[ub * 4 + ty];
		const int ce = (cnt_ty < 2 ? 0 : cnt_ty - 1) + 1;  // edges in the list (as hv_raw_kernel counts them)
		const int i1 = min(i0 + RAW_T - 1, u.L1 - 1);
		const int smp0 = (int)(i0 * (a.fs_d * 1e-3)), smp1 = (int)(i1 * (a.fs_d * 1e-3));
		const int q0 = min(a.n_tiles, max(0, smp0 / SD_CH));
		const int q1 = min(a.n_tiles, smp1 / SD_CH + 1);
		auto T = [&](int c) { return trun[c * 4 + ty]; };
		const int base = max(0, min(T(q0), ce) - 4);
		const int end = min(ce, min(T(q1), ce) + 4);
		int c = q0;
		while (c > 0 && T(c) > base) --c;  // first chunk that holds an edge of the slice
		const int t0 = T(c), t1 = T(min(c + 1, a.n_tiles)), t2 = T(min(c + 2, a.n_tiles)), t3 = T(min(c + 3, a.n_tiles)),
				  t4 = T(min(c + 4, a.n_tiles));
		const int mode = (end - base - 1 > RAW_LDS) ? 1 : ((end <= t4 || c + 4 >= a.n_tiles) ? 0 : 2);
		d0 = make_int4(base, end, c, mode);
		d1 = make_int4(t0, t1, t2, t3);
	}
	desc[2 * gid] = d0;
	desc[2 * gid + 1] = d1;
}

template <bool SLOTS, bool DESC>
__global__ __launch_bounds__(RAW_T) void hv_raw_kernel(RawArgs a) {
	__shared__ double X[4][RAW_LDS], Y[4][RAW_LDS + 1];  // (Y holds the slice's edges first: interval j replaces edge j once both of its edges are read)
	__shared__ int s_base[4], s_len[4];
	__shared__ __attribute__((aligned(8))) unsigned short LO[4][RAW_T];  // per type and frame of the block: how many staged intervals lie at or before the frame
	const int tid = threadIdx.x, lane = tid & 63, ty_w = tid >> 6;
#if WC_RQ_PROF
	long long acc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
	long long last_ = clock64();
#endif
	const int band = blockIdx.y;
	const HvUtt u = a.utts[blockIdx.z];
	const int i0 = blockIdx.x * RAW_T;
	if (i0 >= u.L1) return;
	const int i = i0 + tid;
	const int *cnt = a.ev_count + ((long long)blockIdx.z * a.n_bands + band) * 4;
	const int cap = a.ev_cap[band];
	const double fb = a.band_f0[band];  // (asked for up front: behind the frames it was a round trip of its own)
	const double *__restrict__ ev = SLOTS ? nullptr : a.events + u.ev_off + a.ev_band_off[band];
	const int scap = SLOTS ? a.slot_cap[band] : 0;
	const double *__restrict__ slot = SLOTS ? a.slots + blockIdx.z * a.slots_per_utt + a.slot_off[band] : nullptr;
	const int *__restrict__ trun_b = a.tile_run + ((long long)blockIdx.z * a.n_bands + band) * (a.n_tiles + 1) * 4;
	// edge q of the list of type ty, wherever it lies
	auto edge = [&](int ty, int q) -> double {
		if (!SLOTS) return ev[(long long)ty * cap + q];
		int lo = 0, hi = a.n_tiles;  // last chunk c with tile_run[c] <= q
		while (hi - lo > 1) {
			const int mid = (lo + hi) >> 1;
			if (trun_b[mid * 4 + ty] <= q) lo = mid; else hi = mid;
		}
		return slot[((long long)lo * 4 + ty) * scap + min(q - trun_b[lo * 4 + ty], scap - 1)];
	};
	double *__restrict__ out = a.raw + u.l1_off * a.n_bands + (long long)band * u.L1;
	// the band-pass kernel recorded how many edges precede every tile, so the slice is bounded without searching: the running
	// counts of the chunks around the block in one load (lane l: chunk q0 - 2 + l), requested before anything depends on them
	const int i1 = min(i0 + RAW_T - 1, u.L1 - 1);
	// (the chunks the block's first and last frames fall into only bound the slice: a sample either way costs nothing -- four
	// edges of margin on both sides, and a frame whose answer is not strictly inside the slice takes the whole list -- so no
	// exact quotients here: two FP64 and two integer divisions per wavefront were an eighth of the kernel's instructions)
	const int smp0 = (int)(i0 * (a.fs_d * 1e-3)), smp1 = (int)(i1 * (a.fs_d * 1e-3));
	const int q0 = min(a.n_tiles, max(0, SLOTS ? smp0 / SD_CH : smp0 / a.tile_adv));
	const int q1 = min(a.n_tiles, (SLOTS ? smp1 / SD_CH : smp1 / a.tile_adv) + 1);
	const int tr_first = q0 - 2;
	const int tr_mine = DESC ? 0 : trun_b[min(max(tr_first + (lane & 7), 0), a.n_tiles) * 4 + ty_w];
	// number of intervals = edges - 1 (0 when fewer than 2 edges); all four need more than 2 (reference :1101-1107)
	int n[4];
	bool ok = true;
#pragma unroll
	for (int ty = 0; ty < 4; ++ty) {
		const int ce = SLOTS ? cnt[ty] : min(cnt[ty], cap);  // (an overflowed list is re-done by the caller; stay inside the buffer)
		n[ty] = ce < 2 ? 0 : ce - 1;
		ok = ok && n[ty] > 2;
	}
	if (!ok) {
		if (i < u.L1) out[i] = 0.0;
		return;
	}
	const double fs = a.fs_d;
	RQ_T(0);
	if (DESC) {
		// the slice as hv_rawdesc_kernel left it: scalar loads, nothing to work out in front of the edges' own loads
		const int ty = __builtin_amdgcn_readfirstlane(ty_w);
		const int4 *__restrict__ dp = a.desc + ((((long long)blockIdx.z * a.n_bands + band) * a.desc_blocks + blockIdx.x) * 4 + ty) * 2;
		const int4 d0 = dp[0], d1 = dp[1];
		const int base = d0.x, end = d0.y, c0 = d0.z, mode = d0.w;
		const int len = end - base - 1;
		RQ_T(1);
		if (lane == 0) { s_base[ty] = (mode != 1) ? base : -1; s_len[ty] = len; }
		if (mode != 1) {
			double *__restrict__ E = Y[ty];
			if (mode == 0) {
				const int t0 = d1.x, t1 = d1.y, t2 = d1.z, t3 = d1.w;
				const double *__restrict__ sl0 = slot + ((long long)c0 * 4 + ty) * scap;  // (the four chunks' slots lie 4 scap apart)
				for (int j = lane; j <= len; j += 64) {  // the slice's edges, then interval j over edge j once both of its edges are read
					const int q = base + j;
					const int kq = (q >= t1 ? 1 : 0) + (q >= t2 ? 1 : 0) + (q >= t3 ? 1 : 0);
					const int tk = kq == 0 ? t0 : (kq == 1 ? t1 : (kq == 2 ? t2 : t3));
					E[j] = sl0[kq * 4 * scap + min(q - tk, scap - 1)];
				}
			} else {
				for (int c = c0; c < a.n_tiles && trun_b[c * 4 + ty] < end; ++c) {
					const int tc = trun_b[c * 4 + ty];
					const int lo = max(base, tc), hi = min(end, trun_b[(c + 1) * 4 + ty]);
					const double *__restrict__ src = slot + ((long long)c * 4 + ty) * scap;
					for (int q = lo + lane; q < hi; q += 64) E[q - base] = src[min(q - tc, scap - 1)];
				}
			}
			rq_fence();
			for (int j0 = 0; j0 < len; j0 += 64) {  // (one wavefront per type: the reads of a trip precede its writes)
				const int j = j0 + lane;
				double ea = 0.0, eb = 1.0;
				if (j < len) { ea = E[j]; eb = E[j + 1]; }
				if (j < len) {
					X[ty][j] = div_const((ea + eb) / 2.0, fs, a.r_fs_d);
					Y[ty][j] = fs / (eb - ea);
				}
			}
			raw_frame_counts(X[ty], LO[ty], len, i0, lane);
		}
	} else {
		const int ty = ty_w;
		const int *__restrict__ trun = trun_b;
		const int ce = n[ty] + 1;  // edges in the list
		auto T = [&](int c) -> int {  // tile_run[c] of this type (c wave-uniform)
			const int l = c - tr_first;
			return (l >= 0 && l < 8) ? __shfl(tr_mine, l, 64) : trun[c * 4 + ty];
		};
		const int base = max(0, min(T(q0), ce) - 4);
		const int end = min(ce, min(T(q1), ce) + 4);
		const int len = end - base - 1;  // staged intervals base .. base + len - 1
		RQ_T(1);
		if (lane == 0) { s_base[ty] = (len <= RAW_LDS) ? base : -1; s_len[ty] = len; }
		if (len <= RAW_LDS) {
			double *__restrict__ E = Y[ty];  // edges base .. end - 1 (general path)
			bool done = false;
			if (SLOTS) {
				int c = q0;
				while (c > 0 && T(c) > base) --c;  // first chunk that holds an edge of the slice
				const int t0 = T(c), t1 = T(min(c + 1, a.n_tiles)), t2 = T(min(c + 2, a.n_tiles)), t3 = T(min(c + 3, a.n_tiles)),
						  t4 = T(min(c + 4, a.n_tiles));
				if (end <= t4 || c + 4 >= a.n_tiles) {
					// the usual case, a slice within four chunks (the block's own chunk, a few edges either side): both edges of an
					// interval straight from their slots
					auto at = [&](int q) -> double {
						const int k = (q >= t1 ? 1 : 0) + (q >= t2 ? 1 : 0) + (q >= t3 ? 1 : 0);
						const int tk = k == 0 ? t0 : (k == 1 ? t1 : (k == 2 ? t2 : t3));
						return slot[((long long)(c + k) * 4 + ty) * scap + min(q - tk, scap - 1)];
					};
					for (int j = lane; j < len; j += 64) {
						const double ea = at(base + j), eb = at(base + j + 1);
						X[ty][j] = div_const((ea + eb) / 2.0, fs, a.r_fs_d);
						Y[ty][j] = fs / (eb - ea);
					}
					done = true;
				} else {
					for (; c < a.n_tiles && T(c) < end; ++c) {
						const int tc = T(c);
						const int lo = max(base, tc), hi = min(end, T(c + 1));
						const double *__restrict__ src = slot + ((long long)c * 4 + ty) * scap;
						for (int q = lo + lane; q < hi; q += 64) E[q - base] = src[min(q - tc, scap - 1)];
					}
				}
			} else {
				const double *__restrict__ e = ev + (long long)ty * cap;
				for (int j = lane; j < len; j += 64) {
					const double ea = e[base + j], eb = e[base + j + 1];
					X[ty][j] = div_const((ea + eb) / 2.0, fs, a.r_fs_d);
					Y[ty][j] = fs / (eb - ea);
				}
				done = true;
			}
			if (!done)
				for (int j0 = 0; j0 < len; j0 += 64) {  // (one wavefront per type: the reads of a trip precede its writes)
					const int j = j0 + lane;
					double ea = 0.0, eb = 1.0;
					if (j < len) { ea = E[j]; eb = E[j + 1]; }
					if (j < len) {
						X[ty][j] = div_const((ea + eb) / 2.0, fs, a.r_fs_d);
						Y[ty][j] = fs / (eb - ea);
					}
				}
			raw_frame_counts(X[ty], LO[ty], len, i0, lane);
		}
	}
	RQ_T(2);
	__syncthreads();
	RQ_T(3);
	if (i >= u.L1) return;
	const double t = div_const((double)(i * 1), 1000.0, 1.0 / 1000.0);  // i * 1 / 1000.0
	double s = 0.0;
#pragma unroll
	for (int ty = 0; ty < 4; ++ty) {  // (a + b + c + d) in the reference's order: negative-going, positive-going, peaks, dips
		const int base = s_base[ty];
		double x0, x1, y0, y1;
		bool staged = false;
		if (base >= 0) {
			// c = #{k < n : X[k] <= t} within the staged intervals, valid only if the answer is strictly inside them
			// (otherwise the true boundary may lie outside: fall back to the whole list)
			const double *xs = X[ty];
			const int len = s_len[ty];
			const int lo = LO[ty][tid];  // #{j < len : X[j] <= t}
			const int c = base + lo;
			if ((lo > 0 || base == 0) && (lo < len || base + len == n[ty])) {
				const int k = min(max(c, 1), n[ty] - 1) - base;  // interp1 between intervals k - 1 and k
				if (k >= 1 && k < len) {
					x0 = xs[k - 1]; x1 = xs[k];
					y0 = Y[ty][k - 1]; y1 = Y[ty][k];
					staged = true;
				}
			}
		}
		if (!staged) {
			auto eg = [&](int q) { return edge(ty, q); };
			const int c = hv_count_le(eg, 0, n[ty], fs, t);
			const int k = min(max(c, 1), n[ty] - 1);
			const double e0 = eg(k - 1), e1 = eg(k), e2 = eg(k + 1);
			x0 = div_const((e0 + e1) / 2.0, fs, a.r_fs_d); x1 = div_const((e1 + e2) / 2.0, fs, a.r_fs_d);
			y0 = fs / (e1 - e0); y1 = fs / (e2 - e1);
		}
		// interp1 (reference src/world_matlabfunctions.cpp:157-182) of the intervals fs / (e[k+1] - e[k]) located
		// at the midpoints (reference src/harvest.cpp:1210-1213)
		const double sl = (t - x0) / (x1 - x0);
		const double v = y0 + sl * (y1 - y0);
		s = (ty == 0) ? v : s + v;
	}
	RQ_T(4);
	double v = s / 4.0;
	if (v > fb * 1.1 || v < fb * 0.9 || v > a.f0_ceil || v < a.f0_floor) v = 0.0;
	out[i] = v;
	RQ_T(5);
	RQ_FLUSH();
}

// A frame whose interval does not lie strictly inside the staged slice (or whose slice was too long to stage) takes its three edges
// from the slots themselves: a bisection of the running counts per edge.  Rare, and long: kept out of line, so that the sixteen places
// of hv_raw_wave_kernel that may need it share one copy (inlined, they made 90 KB of code).
struct RawQuad { double x0, x1, y0, y1; };
__device__ __attribute__((noinline)) RawQuad raw_fallback(const double *slot, const int *trun_b, int scap, int n_tiles, double fs, double r_fs_d, int ty,
														  int n_ty, double t) {
	auto eg = [&](int q) -> double {
		int lo = 0, hi = n_tiles;
		while (hi - lo > 1) {
			const int mid = (lo + hi) >> 1;
			if (trun_b[mid * 4 + ty] <= q) lo = mid; else hi = mid;
		}
		return slot[((long long)lo * 4 + ty) * scap + min(q - trun_b[lo * 4 + ty], scap - 1)];
	};
	const int c = hv_count_le(eg, 0, n_ty, fs, t);
	const int k = min(max(c, 1), n_ty - 1);
	const double e0 = eg(k - 1), e1 = eg(k), e2 = eg(k + 1);
	RawQuad r;
	r.x0 = div_const((e0 + e1) / 2.0, fs, r_fs_d); r.x1 = div_const((e1 + e2) / 2.0, fs, r_fs_d);
	r.y0 = fs / (e1 - e0); r.y1 = fs / (e2 - e1);
	return r;
}

// The same block of 256 frames by ONE wavefront (round 6, default for the sliding band-pass's slots).  Above, four wavefronts stage a
// type each, meet at a barrier, and a thread per frame reads all four: a chain of scalar loads, edge loads, LDS round trips and the
// barrier per block with ~300 instructions per wavefront to show for it -- 0.52 of the kernel's time is issue, seven blocks per
// CU (19.5 KB of LDS each) do not cover the chain.  Here a wavefront takes the four types in turn through ONE 4.9 KB buffer: the
// next type's edges are requested before the current type is worked on (registers), no barrier, a lane owns frames 4 lane .. 4 lane
// + 3 (their counts come out of raw_frame_counts' scan in registers) and adds the types up in the reference's order.  The same
// arithmetic per (frame, type): the same bits.
__device__ __forceinline__ void raw_frame_counts4(const double *__restrict__ Xs, unsigned short *__restrict__ lo_, int len, int i0, int lane, int (&m)[4]) {
	*reinterpret_cast<uint2 *>(&lo_[4 * lane]) = make_uint2(0u, 0u);
	rq_fence();
	for (int j0 = 0; j0 < len; j0 += 64) {
		const int j = j0 + lane;
		int rel = RAW_T;
		if (j < len) {
			const double x = Xs[j];
			int f = (int)ceil(x * 1000.0);
			while (f > 0 && div_const((double)(f - 1), 1000.0, 1.0 / 1000.0) >= x) --f;   // the first frame with f / 1000.0 >= x, exactly
			while (div_const((double)f, 1000.0, 1.0 / 1000.0) < x) ++f;
			rel = max(f - i0, 0);
		}
		bool pending = rel < RAW_T;
		while (__ballot(pending) != 0ull) {
			if (pending) lo_[rel] = (unsigned short)(j + 1);
			rq_fence();
			pending = pending && lo_[rel] < j + 1;
		}
	}
	rq_fence();
	const uint2 w2 = *reinterpret_cast<const uint2 *>(&lo_[4 * lane]);
	int m0 = (int)(w2.x & 0xFFFFu), m1 = (int)(w2.x >> 16), m2 = (int)(w2.y & 0xFFFFu), m3 = (int)(w2.y >> 16);
	m1 = max(m1, m0); m2 = max(m2, m1); m3 = max(m3, m2);
	int inc = m3;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const int t = __shfl_up(inc, o, 64);
		if (lane >= o) inc = max(inc, t);
	}
	int ex = __shfl_up(inc, 1, 64);
	if (lane == 0) ex = 0;
	m[0] = max(m0, ex); m[1] = max(m1, ex); m[2] = max(m2, ex); m[3] = max(m3, ex);
}

__global__ __launch_bounds__(64) void hv_raw_wave_kernel(RawArgs a) {
	__shared__ double X[RAW_LDS], Y[RAW_LDS + 1];  // (Y holds the slice's edges first, as above)
	__shared__ __attribute__((aligned(8))) unsigned short LO[RAW_T];
	const int lane = threadIdx.x;
	const int band = blockIdx.y;
	const HvUtt u = a.utts[blockIdx.z];
	const int i0 = blockIdx.x * RAW_T;
	if (i0 >= u.L1) return;
	const int *cnt = a.ev_count + ((long long)blockIdx.z * a.n_bands + band) * 4;
	const double fb = a.band_f0[band];
	const int scap = a.slot_cap[band];
	const double *__restrict__ slot = a.slots + blockIdx.z * a.slots_per_utt + a.slot_off[band];
	const int *__restrict__ trun_b = a.tile_run + ((long long)blockIdx.z * a.n_bands + band) * (a.n_tiles + 1) * 4;
	double *__restrict__ out = a.raw + u.l1_off * a.n_bands + (long long)band * u.L1;
	const int fi = i0 + 4 * lane;  // this lane's first frame
	int n[4];
	bool ok = true;
#pragma unroll
	for (int ty = 0; ty < 4; ++ty) {
		const int ce = cnt[ty];
		n[ty] = ce < 2 ? 0 : ce - 1;
		ok = ok && n[ty] > 2;
	}
	if (!ok) {
#pragma unroll
		for (int m = 0; m < 4; ++m) if (fi + m < u.L1) out[fi + m] = 0.0;
		return;
	}
	const double fs = a.fs_d;
	const int4 *__restrict__ dp = a.desc + ((((long long)blockIdx.z * a.n_bands + band) * a.desc_blocks + blockIdx.x) * 4) * 2;
	int4 d0[4], d1[4];
#pragma unroll
	for (int ty = 0; ty < 4; ++ty) { d0[ty] = dp[2 * ty]; d1[ty] = dp[2 * ty + 1]; }
	// the edges of a slice within four chunks (mode 0), elements lane + 64 k <= len <= RAW_LDS, straight from their slots
	auto fetch = [&](int ty, double (&pe)[5]) {
		const int base = d0[ty].x, len = d0[ty].y - base - 1, c0 = d0[ty].z;
		const int t0 = d1[ty].x, t1 = d1[ty].y, t2 = d1[ty].z, t3 = d1[ty].w;
		const double *__restrict__ sl0 = slot + ((long long)c0 * 4 + ty) * scap;
#pragma unroll
		for (int k = 0; k < 5; ++k) {
			const int j = lane + 64 * k;
			pe[k] = 0.0;
			if (d0[ty].w == 0 && j <= len) {
				const int q = base + j;
				const int kq = (q >= t1 ? 1 : 0) + (q >= t2 ? 1 : 0) + (q >= t3 ? 1 : 0);
				const int tk = kq == 0 ? t0 : (kq == 1 ? t1 : (kq == 2 ? t2 : t3));
				pe[k] = sl0[kq * 4 * scap + min(q - tk, scap - 1)];
			}
		}
	};
	static_assert(RAW_LDS < 5 * 64, "five edges per lane cover a staged slice");
	double tt[4], s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
	for (int m = 0; m < 4; ++m) tt[m] = div_const((double)((fi + m) * 1), 1000.0, 1.0 / 1000.0);  // i * 1 / 1000.0
	double pe[5];
	fetch(0, pe);
#pragma unroll
	for (int ty = 0; ty < 4; ++ty) {
		const int base = d0[ty].x, end = d0[ty].y, c0 = d0[ty].z, mode = d0[ty].w;
		const int len = end - base - 1;
		const bool staged_ty = mode != 1;
		int lo4[4] = {0, 0, 0, 0};
		if (staged_ty) {
			double *__restrict__ E = Y;
			if (mode == 0) {
#pragma unroll
				for (int k = 0; k < 5; ++k) if (lane + 64 * k <= len) E[lane + 64 * k] = pe[k];
			} else {
				for (int c = c0; c < a.n_tiles && trun_b[c * 4 + ty] < end; ++c) {
					const int tc = trun_b[c * 4 + ty];
					const int lo = max(base, tc), hi = min(end, trun_b[(c + 1) * 4 + ty]);
					const double *__restrict__ src = slot + ((long long)c * 4 + ty) * scap;
					for (int q = lo + lane; q < hi; q += 64) E[q - base] = src[min(q - tc, scap - 1)];
				}
			}
		}
		if (ty < 3) fetch(ty + 1, pe);  // (in flight while this type is worked on)
		if (staged_ty) {
			rq_fence();
			for (int j0 = 0; j0 < len; j0 += 64) {  // (the reads of a trip precede its writes)
				const int j = j0 + lane;
				double ea = 0.0, eb = 1.0;
				if (j < len) { ea = Y[j]; eb = Y[j + 1]; }
				if (j < len) {
					X[j] = div_const((ea + eb) / 2.0, fs, a.r_fs_d);
					Y[j] = fs / (eb - ea);
				}
			}
			raw_frame_counts4(X, LO, len, i0, lane, lo4);
		}
#pragma unroll
		for (int m = 0; m < 4; ++m) {
			const int i = fi + m;
			if (i >= u.L1) continue;
			const double t = tt[m];
			double x0, x1, y0, y1;
			bool staged = false;
			if (staged_ty) {
				const int lo = lo4[m];  // #{j < len : X[j] <= t}
				const int c = base + lo;
				if ((lo > 0 || base == 0) && (lo < len || base + len == n[ty])) {
					const int k = min(max(c, 1), n[ty] - 1) - base;  // interp1 between intervals k - 1 and k
					if (k >= 1 && k < len) {
						x0 = X[k - 1]; x1 = X[k];
						y0 = Y[k - 1]; y1 = Y[k];
						staged = true;
					}
				}
			}
			if (!staged) {
				const RawQuad r = raw_fallback(slot, trun_b, scap, a.n_tiles, fs, a.r_fs_d, ty, n[ty], t);
				x0 = r.x0; x1 = r.x1; y0 = r.y0; y1 = r.y1;
			}
			const double sl = (t - x0) / (x1 - x0);
			const double v = y0 + sl * (y1 - y0);
			s[m] = (ty == 0) ? v : s[m] + v;  // (a + b + c + d) in the reference's order: negative-going, positive-going, peaks, dips
		}
		rq_fence();  // (the buffer is the next type's)
	}
#pragma unroll
	for (int m = 0; m < 4; ++m) {
		const int i = fi + m;
		if (i >= u.L1) continue;
		double v = s[m] / 4.0;
		if (v > fb * 1.1 || v < fb * 0.9 || v > a.f0_ceil || v < a.f0_floor) v = 0.0;
		out[i] = v;
	}
}

// ------------------------------------------------------------------------------------------------
// per-frame candidate detection (reference :1005-1083)
// ------------------------------------------------------------------------------------------------
__global__ void hv_detect_kernel(const HvUtt *__restrict__ utts, const double *__restrict__ raw, double *__restrict__ cand0,
								 int n_bands, int S) {
	const HvUtt u = utts[blockIdx.y];
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= u.L1) return;
	const double *__restrict__ r = raw + u.l1_off * n_bands + i;
	double *__restrict__ out = cand0 + (u.l1_off + i) * S;
	int nc = 0, st = 0;
	int prev = 0;
	double sum = 0.0;
	constexpr int B = 8;  // bands whose values are requested together (a load per band and thread, each a round trip if left in the loop's order)
	for (int j0 = 1; j0 < n_bands; j0 += B) {
		double vv[B];
#pragma unroll
		for (int b = 0; b < B; ++b) vv[b] = r[(long long)min(j0 + b, n_bands - 1) * u.L1];
#pragma unroll
		for (int b = 0; b < B; ++b) {
			const int j = j0 + b;
			if (j >= n_bands) break;
			const double v = vv[b];
			const int cur = (j == n_bands - 1) ? 0 : (v > 0 ? 1 : 0);
			if (cur - prev == 1) { st = j; sum = 0.0; }
			if (cur - prev == -1) {
				const int ed = j;
				if (ed - st >= 10 && nc < S) out[nc++] = sum / (ed - st);
			}
			if (cur) sum += v;
			prev = cur;
		}
	}
	for (int k = nc; k < S; ++k) out[k] = 0.0;
}

// ------------------------------------------------------------------------------------------------
// refinement by instantaneous frequency (reference :750-982), one wavefront per candidate
// ------------------------------------------------------------------------------------------------
struct RefArgs {
	const HvUtt *utts;
	int n_utt;
	const double *y;
	const double *cand0;
	const double2 *tw;
	const double2 *rot;  // per half window length hw: (cos, sin) of 2 pi / (2 hw + 1) and of 8 times that
	const double2 *rot8; // per half window length hw: (cos, sin) of k 2 pi / (2 hw + 1), k = 0 .. 7
	const double *cos_table;  // HarvestOption::use_cos_table: the reference's 8001-entry cosine table (src/harvest.cpp:152-170)
	double *cand1, *score1;
	long long total_frames;
	int max_l1;  // frames of the longest utterance (grid of hv_refine_group_kernel)
	HvParams p;
	int *flags;  // [0]: a rate-bounded buffer overflowed; [1]: a raw candidate sits on a tie of the refinement's integer decisions (below); [2 + u]: ... in utterance u
};

#ifndef WC_REFINE_WAVES
#define WC_REFINE_WAVES 4
#endif
#ifndef WC_REFINE_FENCE
#define WC_REFINE_FENCE 1
#endif
constexpr int RF_MAXHW = 1023;          // longest half window: 2 hw + 1 < 2048 keeps the transform size of reference :962 within the 4096-entry twiddle table
                                        // (f0 = 37.5 Hz at 8 kHz needs 321; 16 kHz after decimation and a 47 Hz candidate 511)
constexpr int RF_MAXW = 2 * RF_MAXHW + 1;

// The window phase at a lane's first sample n = sub (reference :762-788) is the phase at sample 0 -- the reference's expression, one
// sincos per (frame, window) -- turned by sub steps of beta = 2 pi / (2 hw + 1) out of a table (round 6: the sincos of a key is
// evaluated once per key instead of once per lane and pass; hv_refine_group_kernel does it in front of the passes.  All three
// kernels form it this way, so they agree bit for bit)
__device__ __forceinline__ double2 rf_phase0(int hw, int basic, double pos, double fs) {
	const double wlt = (2.0 * hw + 1.0) / fs;
	const double tmp = (basic - 1.0) / fs - pos;
	const double tmp2 = 2.0 * kPi * tmp / wlt;
	double c, s;
	sincos(tmp2, &s, &c);
	return make_double2(c, s);
}
__device__ __forceinline__ double2 rf_turn(double2 p, double2 r) {
	return make_double2(fma(p.x, r.x, -(p.y * r.y)), fma(p.y, r.x, p.x * r.y));
}

// One workgroup per 1 ms frame; a wavefront handles candidate slot j of all 7 overlap blocks at once:
// lanes 8 b .. 8 b + 7 work on block b (the same slot of frames i-3 .. i+3, so the windows have similar
// lengths) and split the window samples n = sub + 8 q among them.  The Blackman window and the DFT twiddles
// advance by rotation recurrences from exact starting values (one sincos and 6 table twiddles per lane);
// the 24 partial sums are reduced inside the 8-lane group by a halving butterfly that leaves harmonic h
// in lane h.
// TABLE: HarvestOption::use_cos_table -- the main window from the reference's cosine table with its own index arithmetic
// (getMainWindow, src/harvest.cpp:779-787), the difference window from its neighbours (:792-803); three look-up pairs per
// sample instead of a rotation, an option nobody takes for speed here but whose results differ from exact cosines by 1e-4.
template <bool TABLE>
__global__ __launch_bounds__(256, WC_REFINE_WAVES) void hv_refine_kernel(RefArgs a) {  // 4 waves per SIMD (128 VGPRs, 68 bytes of scratch): 9.5 ms against 9.9 ms at 3 waves / 156 VGPRs since the per-lane loop
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int blk = lane >> 3, sub = lane & 7;
	const long long g = blockIdx.x;
	if (g >= a.total_frames) return;
	const int ui = hv_find(a.utts, a.n_utt, g, &HvUtt::l1_off);
	const HvUtt u = a.utts[ui];
	const int i = (int)(g - u.l1_off);
	const double pos = i * 1 / 1000.0;
	const double fs = a.p.fs_d;
	const int S = a.p.S;
	const double *__restrict__ y = a.y + u.y_off;
	const int src = (blk == 0) ? i : (blk <= 3 ? i - blk : i + (blk - 3));  // overlap (reference :987-1000)
	const bool src_ok = blk < 7 && src >= 0 && src < u.L1;
	const double *__restrict__ crow = a.cand0 + (u.l1_off + i) * S;  // wave-uniform row of this frame; the lane's source frame is doff away
	const int doff = (src - i) * S;
	// Phase 1: the window phase at every lane's first sample, exactly as the reference evaluates it (:762-788), for all the
	// candidates this wavefront will take (slots wv, wv + 4, ...), parked in LDS.  Kept apart from the recurrences of phase 2
	// because the polynomial coefficients of sincos are loop invariant: inside the candidate loop the compiler holds their 18
	// registers across the whole body and, at 128 registers, spills them -- 84 bytes of scratch per thread, written by every
	// one of the 164 M threads of a batch (7 GB of HBM writes per step in round 1's profile).
	__shared__ double2 start_phase[MAX_SLOTS / 4][256];
	__shared__ double row_f[7 * MAX_SLOTS], row_s[7 * MAX_SLOTS];
	for (int j = wv, q = 0; j < S; j += 4, ++q) {
		double f = 0.0;
		if (src_ok) f = crow[doff + j];
		if (__ballot(f > 0.0) == 0ull) continue;  // an empty slot (most are) is skipped in phase 2 as well
		const double fc = f > 0.0 ? f : 100.0;
		const int hw = min((int)(1.5 * fs / fc + 1.0), RF_MAXHW);
		const double bt0 = (-hw) / fs;
		const int basic = mround((pos + bt0) * fs + 0.001);
		start_phase[q][threadIdx.x] = rf_turn(rf_phase0(hw, basic, pos, fs), a.rot8[hw * 8 + sub]);
	}
	for (int j = wv, q = 0; j < S; j += 4, ++q) {
		double f = 0.0;
		if (src_ok) f = crow[doff + j];
		const bool live = f > 0.0;
		double rf = 0.0, rs = 0.0;
		if (__ballot(live) != 0ull) {
			// per-candidate constants (garbage-free defaults for idle groups)
			const double fc = live ? f : 100.0;
			const int hw = min((int)(1.5 * fs / fc + 1.0), RF_MAXHW);
			const int bt = live ? 2 * hw + 1 : 0;
			const double wlt = (2.0 * hw + 1.0) / fs;
			// 2 + int(log(2 hw + 1) / log 2) of reference :962: 2 hw + 1 is odd, so the logarithm is never close to an
			// integer and the floor is the position of the leading bit
			const int fft_index = 2 + (31 - __clz(hw * 2 + 1));
			const int N = 1 << fft_index;
			const int tsh = kTwiddleN / N;
			const double bt0 = (-hw) / fs;
			const int basic = mround((pos + bt0) * fs + 0.001);
			const int nh = min((int)(fs / 2.0 / fc), 6);
			// harmonic bins (reference :853-861).  Formed twice -- here for the recurrence coefficients, again behind the sample loop
			// for the closing twiddles -- because six registers held across the loop are six too many at 128
			double bin_unit = fc * N / fs;
			auto bin_of = [&](int h) { return mround(bin_unit * (h + 1)); };
			double wc = start_phase[q][threadIdx.x].x, ws = start_phase[q][threadIdx.x].y;  // phase 1
			const double2 r1 = a.rot[2 * hw], r8 = a.rot[2 * hw + 1];  // (cos, sin) of beta and 8 beta, beta = 2 pi / (2 hw + 1)
			const double k1 = 0.5 * r1.y, k2 = 0.16 * (2.0 * r1.y * r1.x);
			// DFT bins of the lane's sub-sequence n = sub + 8 q by Goertzel's recurrence s_q = x_q + 2 cos(w) s_{q-1} - s_{q-2}
			// (w = 8 phi_h): two instructions per sample, harmonic and window instead of the four plus a twiddle rotation
			// of a direct accumulation.  sum_q x_q e^{-i w q} = e^{-i w (Q-1)} (s_{Q-1} - e^{-i w} s_{Q-2}); with the lane's
			// own phase e^{-i phi sub} both factors are exact table twiddles.  The loop is unrolled by two so the
			// (s_{q-1}, s_{q-2}) pair alternates between two register sets; trailing zero samples are harmless.
			double c2[6];
#pragma unroll
			for (int h = 0; h < 6; ++h) c2[h] = 2.0 * a.tw[((bin_of(h) * 8) & (N - 1)) * tsh].x;
			double sa[12], sb[12];  // [2 h] main window, [2 h + 1] difference window; sa = newest after a full trip
#pragma unroll
			for (int k = 0; k < 12; ++k) { sa[k] = 0.0; sb[k] = 0.0; }
			auto table_window = [&](int n) -> double {  // reference :779-787, operation by operation
				const double two_pi = 2.0 * kPi;
				const double tmp = (basic + n - 1.0) / fs - pos;
				const double tmp2 = two_pi * (tmp / wlt + 1);
				const double dindex = fmod(tmp2, two_pi) / two_pi * 8000;
				const double dindex2 = fmod(dindex * 2, 8000.0);
				return 0.42 + 0.5 * a.cos_table[(int)round(dindex)] + 0.08 * a.cos_table[(int)round(dindex2)];
			};
			auto sample = [&](int n, double &xm, double &xd) {
				const double yv = (n < bt) ? y[clampi(basic + n - 1, 0, u.y_len - 1)] : 0.0;
				if (TABLE) {
					double m = 0.0, d = 0.0;
					if (n < bt) {
						m = table_window(n);
						if (n == 0) d = -table_window(1) / 2.0;
						else if (n == bt - 1) d = table_window(bt - 2) / 2.0;
						else d = -(table_window(n + 1) - table_window(n - 1)) / 2.0;
					}
					xm = m * yv;
					xd = d * yv;
					return;
				}
				// Blackman window 0.42 + 0.5 cos + 0.08 cos 2theta = 0.34 + c (0.5 + 0.16 c); its centred difference
				// -(w[n+1] - w[n-1]) / 2 = sin theta (0.5 sin beta + 0.16 sin 2beta cos theta) in the interior, the
				// one-sided forms of reference :797-798 at the two ends
				const double m = fma(wc, fma(0.16, wc, 0.5), 0.34);
				const bool first = n == 0, last = n == bt - 1;
				const double cnb = fma(wc, r1.x, first ? -(ws * r1.y) : ws * r1.y);  // cos(theta +- beta)
				const double mnb = fma(cnb, fma(0.16, cnb, 0.5), 0.34);
				double d = ws * fma(k2, wc, k1);
				d = first ? -mnb / 2.0 : (last ? mnb / 2.0 : d);
				xm = m * yv;
				xd = d * yv;
				const double nc_ = fma(wc, r8.x, -(ws * r8.y));
				ws = fma(ws, r8.x, wc * r8.y);
				wc = nc_;
			};
			// A lane stops after its own last sample (at most one trailing zero): the refined F0 must be a function of
			// the candidate's window and bins alone.  The reference returns bitwise EQUAL values for candidates of a frame
			// that share them (the window length is quantised, :950-958) and mergeF0's searchScore compares with ==
			// (:463-470); running every lane to the longest window in the wavefront made the result depend on the
			// neighbours through the number of trailing rotations.
			// Interior samples take the short form of the same arithmetic (no one-sided differences, no range checks): only
			// the first trip (sample 0) and a lane's last trip can hold an end of the window, and the lanes of a wavefront
			// reach their last trips together, give or take one (the seven blocks hold the same slot of neighbouring frames).
			auto interior = [&](int n, double &xm, double &xd) {
				const double yv = y[clampi(basic + n - 1, 0, u.y_len - 1)];
				const double m = fma(wc, fma(0.16, wc, 0.5), 0.34);
				const double d = ws * fma(k2, wc, k1);
				xm = m * yv;
				xd = d * yv;
				const double nc_ = fma(wc, r8.x, -(ws * r8.y));
				ws = fma(ws, r8.x, wc * r8.y);
				wc = nc_;
			};
			int Q = 0;
			for (int n = sub; n < bt; n += 16) {
				const bool ends = TABLE || n == sub || n + 9 >= bt;
				const bool any_end = __ballot(ends) != 0ull;  // wave-uniform: the recurrences below stay outside the branch
				double xm, xd;
				if (any_end) sample(n, xm, xd);
				else interior(n, xm, xd);
#pragma unroll
				for (int h = 0; h < 6; ++h) {
					sb[2 * h] = fma(c2[h], sa[2 * h], xm) - sb[2 * h];
					sb[2 * h + 1] = fma(c2[h], sa[2 * h + 1], xd) - sb[2 * h + 1];
				}
				if (any_end) sample(n + 8, xm, xd);
				else interior(n + 8, xm, xd);
#pragma unroll
				for (int h = 0; h < 6; ++h) {
					sa[2 * h] = fma(c2[h], sb[2 * h], xm) - sa[2 * h];
					sa[2 * h + 1] = fma(c2[h], sb[2 * h + 1], xd) - sa[2 * h + 1];
				}
				Q += 2;
			}
			// The 24 partial sums (6 harmonics x {re, im} x {main, difference window}) are reduced inside the 8-lane group by a
			// halving butterfly that leaves harmonic h in lane h.  Its first stage pairs harmonic p with harmonic p + 4 (nothing
			// for p = 2, 3) and is fused with the closing twiddles pair by pair: the recurrence states of a harmonic die as soon
			// as its pair has been exchanged, which keeps the kernel inside 128 registers without scratch.
			asm volatile("" : "+v"(bin_unit));  // (keeps the bins from being carried across the loop instead of re-formed)
			int idx[6];
#pragma unroll
			for (int h = 0; h < 6; ++h) idx[h] = bin_of(h);
			double v[16];
			auto closing = [&](int h, double (&o)[4]) {
				const double2 e1 = a.tw[((idx[h] * (sub + 8 * (Q - 1))) & (N - 1)) * tsh];  // conj = e^{-i phi (sub + 8 (Q-1))}
				const double2 e2 = a.tw[((idx[h] * (sub + 8 * Q)) & (N - 1)) * tsh];
				o[0] = sa[2 * h] * e1.x - sb[2 * h] * e2.x;
				o[1] = sb[2 * h] * e2.y - sa[2 * h] * e1.y;
				o[2] = sa[2 * h + 1] * e1.x - sb[2 * h + 1] * e2.x;
				o[3] = sb[2 * h + 1] * e2.y - sa[2 * h + 1] * e1.y;
			};
#pragma unroll
			for (int p = 0; p < 4; ++p) {
				double lo[4], hi[4] = {0.0, 0.0, 0.0, 0.0};
				closing(p, lo);
				if (p < 2) closing(p + 4, hi);
				const bool up = (sub & 4) != 0;
#pragma unroll
				for (int c = 0; c < 4; ++c) {
					const double send = up ? lo[c] : hi[c];
					const double keep = up ? hi[c] : lo[c];
					v[4 * p + c] = keep + __shfl_xor(send, 4, 64);
				}
#if WC_REFINE_FENCE
				asm volatile("" ::: "memory");  // keeps the next pair's table loads behind this pair's exchange
#endif
			}
#pragma unroll
			for (int k = 0; k < 8; ++k) {
				const bool up = (sub & 2) != 0;
				const double send = up ? v[k] : v[8 + k];
				const double keep = up ? v[8 + k] : v[k];
				v[k] = keep + __shfl_xor(send, 2, 64);
			}
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const bool up = (sub & 1) != 0;
				const double send = up ? v[k] : v[4 + k];
				const double keep = up ? v[4 + k] : v[k];
				v[k] = keep + __shfl_xor(send, 1, 64);
			}
			// lane `sub` = harmonic h: instantaneous frequency and amplitude (fixF0, reference :844-878)
			const int h = sub;
			int myidx = 0;
#pragma unroll
			for (int q = 0; q < 6; ++q) if (q == h) myidx = idx[q];
			const double mr = v[0], mi = v[1], dr = v[2], di = v[3];
			const double pw = mr * mr + mi * mi;
			const double ni = mr * di - mi * dr;
			const double inst = (pw == 0.0) ? 0.0 : (double)myidx * fs / N + ni / pw * fs / 2.0 / kPi;
			const double amp = sqrt(pw);
			const double e_num = amp * inst, e_den = amp * (h + 1.0), e_sc = fabs((inst / (h + 1.0) - fc) / fc);
			double num = 0.0, den = 0.0, sc = 0.0;
#pragma unroll
			for (int q = 0; q < 6; ++q) {  // the reference's summation order over harmonics
				const double x1 = __shfl(e_num, (lane & 56) + q, 64);
				const double x2 = __shfl(e_den, (lane & 56) + q, 64);
				const double x3 = __shfl(e_sc, (lane & 56) + q, 64);
				if (q < nh) { num += x1; den += x2; sc += x3; }
			}
			if (live) {
				rf = num / (den + kSafeH);
				rs = 1.0 / (sc / nh + kSafeH);
				if (rf < a.p.f0_floor || rf > a.p.f0_ceil || rs < 2.5) { rf = 0.0; rs = 0.0; }  // reference :974-979
			}
		}
		if (sub == 0 && blk < 7) {
			row_f[j + S * blk] = rf;
			row_s[j + S * blk] = rs;
		}
	}
	// the frame's 7 S (candidate, score) pairs leave as two contiguous rows instead of 2 x 7 scattered 8-byte stores per
	// wavefront and slot
	__syncthreads();
	for (int k = threadIdx.x; k < 7 * S; k += 256) {
		a.cand1[g * a.p.n_cand + k] = row_f[k];
		a.score1[g * a.p.n_cand + k] = row_s[k];
	}
}

// Exchanges inside a group of eight lanes as DPP moves (VALU, no trip through the LDS crossbar that __shfl_xor's ds_bpermute
// takes): quad permutes for lane ^ 1 and lane ^ 2; lane ^ 4 as a row shift left by four into lanes 0-3 of each eight (banks
// 0 and 2 of the row) and right by four into lanes 4-7 (banks 1 and 3).
#ifndef WC_REFINE_DPP
#define WC_REFINE_DPP 1
#endif
#ifndef WC_REFINE_XCD
#define WC_REFINE_XCD 1
#endif
template <int X>
__device__ __forceinline__ double group8_xor(double v) {
#if WC_REFINE_DPP
	int w[2] = {__double2loint(v), __double2hiint(v)};
#pragma unroll
	for (int k = 0; k < 2; ++k) {
		if (X == 1) w[k] = __builtin_amdgcn_mov_dpp(w[k], 0xB1, 0xF, 0xF, true);       // quad_perm:[1,0,3,2]
		else if (X == 2) w[k] = __builtin_amdgcn_mov_dpp(w[k], 0x4E, 0xF, 0xF, true);  // quad_perm:[2,3,0,1]
		else {
			int r = __builtin_amdgcn_update_dpp(w[k], w[k], 0x104, 0xF, 0x5, false);   // row_shl:4 -> lanes 0-3, 8-11
			w[k] = __builtin_amdgcn_update_dpp(r, w[k], 0x114, 0xF, 0xA, false);       // row_shr:4 -> lanes 4-7, 12-15
		}
	}
	return __hiloint2double(w[1], w[0]);
#else
	return __shfl_xor(v, X, 64);
#endif
}

// The same refinement with the frame's work packed (default; the kernel above is WC_HARVEST_REFINE=slots).
// What a candidate's spectral part -- instantaneous frequency and amplitude of the six harmonics -- depends on is the
// frame, the half window length and the six harmonic bins, not the candidate frequency itself (reference :844-878: only the
// score of :880-893 and the harmonic count use it), and the window length is quantised (:950-958): of the 7 S candidates
// a frame collects from its neighbours about a fifth repeat the (window, bins) key of another one, and most slots
// are empty.  One wavefront per frame therefore
//   1. gathers the live candidates in slot-major order (the same slot of neighbouring frames first: similar windows),
//   2. keeps the first candidate of every distinct key,
//   3. runs the eight-lane machinery of the kernel above on eight such candidates at a time -- all eight groups of the
//      wavefront busy, where a slot of seven overlap blocks filled seven at best and usually fewer,
//   4. scores every other candidate from the harmonics of the one that shares its key, with the same instructions.
// A candidate's refined F0 and score are the values the kernel above computes, bit for bit: the arithmetic of a group of eight
// lanes never depended on what else ran in the wavefront, and candidates with equal keys got equal harmonics there as well.
// 39 % fewer passes through the sample loop and 23 % fewer samples per frame on speech at 48 kHz.
constexpr int RF_GROUP = 2;             // passes whose start phases are staged together (see phase 1 above); a frame's distinct keys rarely need a third pass
// RF_NP: candidate positions of a frame the LDS is sized for (7 S <= RF_NP; 112 covers the reference's default 15 slots, 7 MAX_SLOTS everything)
#ifndef WC_RF_EXP
#define WC_RF_EXP 0  // timing experiments: 1 no sample loop, 2 no closing, 3 gather and de-duplication only
#endif
constexpr int RF_RED_PLANE = 144;       // doubles between the two planes of `red` (1152 bytes: the planes fall into different halves of the banks)
constexpr int RF_RED = RF_RED_PLANE + 128;
// lane i of a row of sixteen receives lane i + Q's value (a DPP row shift: VALU, no trip through the LDS crossbar)
template <int Q>
__device__ __forceinline__ double row_shl_d(double v) {
	const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x100 + Q, 0xF, 0xF, true);
	const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x100 + Q, 0xF, 0xF, true);
	return __hiloint2double(hi, lo);
}
template <bool TABLE, int RF_NP>
__global__ __launch_bounds__(64, RF_NP > 112 ? 2 : WC_REFINE_WAVES) void hv_refine_packed_kernel(RefArgs a) {  // (the LDS of the wide variant allows 11 wavefronts per CU anyway)
	const int lane = threadIdx.x;
	const int grp = lane >> 3, sub = lane & 7;
#if WC_REFINE_XCD
	// frames XCD by XCD: a frame's windows overlap its neighbours', and an eighth of a batch's decimated signals fits one XCD's L2
	const long long g = xcd_frame(blockIdx.x, a.total_frames);
#else
	const long long g = blockIdx.x;
#endif
	if (g >= a.total_frames) return;
	const int ui = hv_find(a.utts, a.n_utt, g, &HvUtt::l1_off);
	const HvUtt u = a.utts[ui];
	const int i = (int)(g - u.l1_off);
	const double pos = i * 1 / 1000.0;
	const double fs = a.p.fs_d;
	const int S = a.p.S;
	const int NC = 7 * S;
	const double *__restrict__ y = a.y + u.y_off;
	const double *__restrict__ crow = a.cand0 + (u.l1_off + i) * S;
	__shared__ double it_f[RF_NP];              // live candidates, slot-major
	__shared__ unsigned long long key[RF_NP];   // half window length and harmonic bins
	__shared__ double row_f[RF_NP], row_s[RF_NP], row_c[RF_NP];  // per position: the three sums over harmonics of :880-893
	__shared__ double2 stage[RF_GROUP][64];     // start phases of a pass; the harmonics it found overwrite them
	__shared__ __attribute__((aligned(16))) double red[RF_RED];  // the eight lanes' closing terms of one harmonic on their way to its sum
	__shared__ int it_basic[RF_NP];             // first sample of the window (reference :762-771)
	__shared__ unsigned char it_pos[RF_NP], it_u[RF_NP], it_rep[RF_NP], un_src[RF_NP], dup_of[RF_NP], it_nh[RF_NP], pos_nh[RF_NP];
	const unsigned long long below = (1ull << lane) - 1ull;

	// 1. live candidates of the overlap (reference :987-1000), slot-major: position k = 7 j + block
	int n = 0;
	for (int base = 0; base < NC; base += 64) {
		const int k = base + lane;
		const int j = k / 7, blk = k - 7 * j;
		const int src = (blk == 0) ? i : (blk <= 3 ? i - blk : i + (blk - 3));
		double f = 0.0;
		if (k < NC && src >= 0 && src < u.L1) f = crow[(src - i) * S + j];
		const bool live = f > 0.0;
		const unsigned long long m = __ballot(live);
		if (k < NC && !live) pos_nh[j + S * blk] = 0;  // an empty position
		if (live) {
			const int at = n + __popcll(m & below);
			const int hw = min((int)(1.5 * fs / f + 1.0), RF_MAXHW);
			const int N = 1 << (2 + (31 - __clz(hw * 2 + 1)));
			const double bin_unit = f * N / fs;
			// key: half window length (11 bits) and the six bins (reference :853-861, :950-962): bin 0 in the upper word, four
			// bits each for what bin h adds to (h + 1) times bin 0 -- rounding keeps that within 3.5
			const int b0 = mround(bin_unit);
			unsigned long long kk = (unsigned long long)hw | ((unsigned long long)(unsigned)b0 << 32);
#pragma unroll
			for (int h = 1; h < 6; ++h) {
				const int dlt = mround(bin_unit * (h + 1)) - (h + 1) * b0 + 8;
				kk |= (unsigned long long)(dlt & 15) << (11 + 4 * (h - 1));
			}
			it_f[at] = f;
			it_pos[at] = (unsigned char)(j + S * blk);
			key[at] = kk;
			// what every lane of a pass needs of the candidate besides its key, worked out once here
			it_basic[at] = mround((pos + (-hw) / fs) * fs + 0.001);
			const int nh = min((int)(fs / 2.0 / f), 6);  // >= 1 below the Nyquist frequency
			it_nh[at] = (unsigned char)nh;
			pos_nh[j + S * blk] = (unsigned char)nh;
			// Ties (round 5).  The window length, the six bins and the number of harmonics are integers cut out of the raw candidate: a
			// candidate within the sliding band-pass's own rounding (1e-14 of it) of one of those cuts gets whichever side that
			// rounding leaves it on -- a train of impulses whose period is a whole number of decimated samples puts 1.5 fs / f0 + 1
			// EXACTLY on an integer in every voiced frame, and the reference's FFT convolution, rounding at 1e-16, decides otherwise
			// than a sliding DFT (profiles/r05_c_impulse_trains.txt: 1.1e-2 Hz and three voicing flips against 1e-11 Hz with the direct
			// FIR).  Such a candidate raises a flag (within 2e-13 relative of a cut: 20 x the rounding, one natural utterance in a
			// thousand); the caller then runs the batch again with the band-pass as a direct FIR sum (hv_exact_twin).
			{
				const double tol = 2e-13;
				const double v = 1.5 * fs / f + 1.0, v2 = fs / 2.0 / f;
				bool tie = fabs(v - rint(v)) < tol * v || (v2 < 7.0 && fabs(v2 - rint(v2)) < tol * v2);
#pragma unroll
				for (int h = 0; h < 6; ++h) {
					const double wv = bin_unit * (h + 1);
					tie = tie || fabs(wv - floor(wv) - 0.5) < tol * wv;
				}
				if (tie) { a.flags[1] = 1; a.flags[2 + ui] = 1; }  // (per utterance since round 6: only the utterances on a tie are run again)
			}
		}
		n += __popcll(m);
	}
	if (n == 0) {  // silence and most unvoiced frames
		for (int k = lane; k < NC; k += 64) {
			a.cand1[g * a.p.n_cand + k] = 0.0;
			a.score1[g * a.p.n_cand + k] = 0.0;
		}
		return;
	}
	// 2. the first candidate of every key, found through a 256-entry table indexed by a hash of the key: the smallest candidate
	//    index that hashes to an entry owns it.  A candidate whose entry belongs to a different key counts as one of a kind and
	//    has its harmonics computed again -- to the same bits; what the table decides is work, never results.
	unsigned int *const tab = reinterpret_cast<unsigned int *>(&stage[0][0]);
	auto slot_of = [](unsigned long long kk) { return (((unsigned)kk * 2654435761u) ^ ((unsigned)(kk >> 32) * 40503u * 65537u)) >> 24; };
	for (int k = lane; k < 256; k += 64) tab[k] = 0xFFFFFFFFu;
	__syncthreads();
	for (int t = lane; t < n; t += 64) atomicMin(&tab[slot_of(key[t])], (unsigned)t);
	__syncthreads();
	int nu = 0, nd = 0;
	for (int t0 = 0; t0 < n; t0 += 64) {
		const int t = t0 + lane;
		int rep = t;
		if (t < n) {
			const unsigned long long mine = key[t];
			const int r = (int)tab[slot_of(mine)];
			if (r != t && key[r] == mine) rep = r;
		}
		const bool uniq = t < n && rep == t, dup = t < n && rep != t;
		const unsigned long long mu = __ballot(uniq), md = __ballot(dup);
		if (uniq) {
			const int r = nu + __popcll(mu & below);
			un_src[r] = (unsigned char)t;
			it_u[t] = (unsigned char)r;
		}
		if (dup) {
			dup_of[nd + __popcll(md & below)] = (unsigned char)t;
			it_rep[t] = (unsigned char)rep;
		}
		nu += __popcll(mu);
		nd += __popcll(md);
	}
	__syncthreads();
	for (int k = lane; k < nd; k += 64) { const int t = dup_of[k]; it_u[t] = it_u[it_rep[t]]; }
	__syncthreads();

	// the sums over harmonics behind a candidate's refined F0 and score, from the harmonics in lanes 0..5 of its group (fixF0,
	// reference :880-893); the quotients of :964-979 wait for the sweep over the row at the end
	auto finish = [&](int ln, double inst, double amp, double fc, int nh, double &num, double &den, double &sc) {
		// valid in lane 0 of every group of eight only (the lane that stores them).  Harmonics beyond nh enter as +0.0: the sums
		// start from +0.0 and can never be -0.0, so adding +0.0 leaves their bits alone.
		const int h = ln & 7;
		const bool on = h < nh;
		const double e_num = on ? amp * inst : 0.0, e_den = on ? amp * (h + 1.0) : 0.0, e_sc = on ? fabs((inst / (h + 1.0) - fc) / fc) : 0.0;
		num = 0.0 + e_num; den = 0.0 + e_den; sc = 0.0 + e_sc;  // the reference's summation order over harmonics
		num += row_shl_d<1>(e_num); den += row_shl_d<1>(e_den); sc += row_shl_d<1>(e_sc);
		num += row_shl_d<2>(e_num); den += row_shl_d<2>(e_den); sc += row_shl_d<2>(e_sc);
		num += row_shl_d<3>(e_num); den += row_shl_d<3>(e_den); sc += row_shl_d<3>(e_sc);
		num += row_shl_d<4>(e_num); den += row_shl_d<4>(e_den); sc += row_shl_d<4>(e_sc);
		num += row_shl_d<5>(e_num); den += row_shl_d<5>(e_den); sc += row_shl_d<5>(e_sc);
	};

	const int npass = WC_RF_EXP == 3 ? 0 : (nu + 7) >> 3;
	for (int c0 = 0; c0 < npass; c0 += RF_GROUP) {
		const int cn = min(RF_GROUP, npass - c0);
		// phase 1 (see the kernel above): window phases at every lane's first sample
		for (int q = 0; q < cn; ++q) {
			const int r = (c0 + q) * 8 + grp;
			const int t_own = r < nu ? un_src[r] : 0;  // (an idle group borrows candidate 0: any valid window will do)
			const int hw = (int)(key[t_own] & 2047ull);
			const int basic = it_basic[t_own];
			stage[q][lane] = rf_turn(rf_phase0(hw, basic, pos, fs), a.rot8[hw * 8 + sub]);
		}
		// phase 2
		for (int q = 0; q < cn; ++q) {
			const int r = (c0 + q) * 8 + grp;
			const bool live = r < nu;
			const int t_own = live ? un_src[r] : 0;
			const unsigned long long kk = key[t_own];
			const int hw = (int)(kk & 2047ull);
			const int bt = (WC_RF_EXP == 1) ? 0 : (live ? 2 * hw + 1 : 0);
			const double wlt = (2.0 * hw + 1.0) / fs;
			const int fft_index = 2 + (31 - __clz(hw * 2 + 1));
			const int N = 1 << fft_index;
			const int basic = it_basic[t_own];
			auto bin_of = [&](int h) -> int {  // harmonic bins (reference :853-861) out of the key
				const int b0 = (int)(kk >> 32);
				return h == 0 ? b0 : (h + 1) * b0 + (int)(((unsigned)kk >> (11 + 4 * (h - 1))) & 15u) - 8;
			};
			double wc = stage[q][lane].x, ws = stage[q][lane].y;
			const double2 r1 = a.rot[2 * hw], r8 = a.rot[2 * hw + 1];
			const double k1 = 0.5 * r1.y, k2 = 0.16 * (2.0 * r1.y * r1.x);
			double c2[6];
#pragma unroll
			for (int h = 0; h < 6; ++h) c2[h] = 2.0 * a.tw[((bin_of(h) * 8) & (N - 1)) << (kTwiddleLog2 - fft_index)].x;
			double sa[12], sb[12];
#pragma unroll
			for (int k = 0; k < 12; ++k) { sa[k] = 0.0; sb[k] = 0.0; }
			auto table_window = [&](int n_) -> double {  // reference :779-787, operation by operation
				const double two_pi = 2.0 * kPi;
				const double tmp = (basic + n_ - 1.0) / fs - pos;
				const double tmp2 = two_pi * (tmp / wlt + 1);
				const double dindex = fmod(tmp2, two_pi) / two_pi * 8000;
				const double dindex2 = fmod(dindex * 2, 8000.0);
				return 0.42 + 0.5 * a.cos_table[(int)round(dindex)] + 0.08 * a.cos_table[(int)round(dindex2)];
			};
			auto sample = [&](int n_, double yl, double &xm, double &xd) {
				const double yv = (n_ < bt) ? yl : 0.0;
				if (TABLE) {
					double m = 0.0, d = 0.0;
					if (n_ < bt) {
						m = table_window(n_);
						if (n_ == 0) d = -table_window(1) / 2.0;
						else if (n_ == bt - 1) d = table_window(bt - 2) / 2.0;
						else d = -(table_window(n_ + 1) - table_window(n_ - 1)) / 2.0;
					}
					xm = m * yv;
					xd = d * yv;
					return;
				}
				const double m = fma(wc, fma(0.16, wc, 0.5), 0.34);
				const bool first = n_ == 0, last = n_ == bt - 1;
				const double cnb = fma(wc, r1.x, first ? -(ws * r1.y) : ws * r1.y);
				const double mnb = fma(cnb, fma(0.16, cnb, 0.5), 0.34);
				double d = ws * fma(k2, wc, k1);
				d = first ? -mnb / 2.0 : (last ? mnb / 2.0 : d);
				xm = m * yv;
				xd = d * yv;
				const double nc_ = fma(wc, r8.x, -(ws * r8.y));
				ws = fma(ws, r8.x, wc * r8.y);
				wc = nc_;
			};
			auto interior = [&](int n_, double yv, double &xm, double &xd) {
				const double m = fma(wc, fma(0.16, wc, 0.5), 0.34);
				const double d = ws * fma(k2, wc, k1);
				xm = m * yv;
				xd = d * yv;
				const double nc_ = fma(wc, r8.x, -(ws * r8.y));
				ws = fma(ws, r8.x, wc * r8.y);
				wc = nc_;
			};
			// the two samples of a trip are loaded one trip ahead (the index is clamped: a load past the window is harmless and unused)
			auto sample_at = [&](int n_) { return y[clampi(basic + n_ - 1, 0, u.y_len - 1)]; };
			double y0 = sample_at(sub), y1 = sample_at(sub + 8);
			int Q = 0;
			for (int n_ = sub; n_ < bt; n_ += 16) {
				const double ny0 = sample_at(n_ + 16), ny1 = sample_at(n_ + 24);
				const bool ends = TABLE || n_ == sub || n_ + 9 >= bt;
				const bool any_end = __ballot(ends) != 0ull;
				double xm, xd;
				if (any_end) sample(n_, y0, xm, xd);
				else interior(n_, y0, xm, xd);
#pragma unroll
				for (int h = 0; h < 6; ++h) {
					sb[2 * h] = fma(c2[h], sa[2 * h], xm) - sb[2 * h];
					sb[2 * h + 1] = fma(c2[h], sa[2 * h + 1], xd) - sb[2 * h + 1];
				}
				if (any_end) sample(n_ + 8, y1, xm, xd);
				else interior(n_ + 8, y1, xm, xd);
#pragma unroll
				for (int h = 0; h < 6; ++h) {
					sa[2 * h] = fma(c2[h], sb[2 * h], xm) - sa[2 * h];
					sa[2 * h + 1] = fma(c2[h], sb[2 * h + 1], xd) - sa[2 * h + 1];
				}
				Q += 2;
				y0 = ny0;
				y1 = ny1;
			}
#if WC_RF_EXP == 2
			{
				double acc = 0.0;
#pragma unroll
				for (int k = 0; k < 12; ++k) acc += sa[k] + sb[k];
				if (acc == 123.456) row_f[lane] = acc;
				continue;
			}
#endif
			// Everything the closing arithmetic needs of the candidate is looked up again from an opaque copy of the lane index: the
			// sample loop above holds 118 registers of recurrence state, and whatever stays live across it is spilled -- once per
			// thread, which at 640 k wavefronts per batch is gigabytes of scratch writes.
			int ln = lane;
			asm volatile("" : "+v"(ln));
			const int sub_c = ln & 7, grp_c = ln >> 3;
			const int r_c = (c0 + q) * 8 + grp_c;
			const bool live_c = r_c < nu;
			const int t_c = live_c ? un_src[r_c] : 0;
			const unsigned long long kk_c = key[t_c];
			const int N_c = 1 << (2 + (31 - __clz((int)(kk_c & 2047ull) * 2 + 1)));
			const int tsh_c = kTwiddleN / N_c;
			int idx[6];
#pragma unroll
			for (int h = 0; h < 6; ++h) {
				const int b0 = (int)(kk_c >> 32);
				idx[h] = h == 0 ? b0 : (h + 1) * b0 + (int)(((unsigned)kk_c >> (11 + 4 * (h - 1))) & 15u) - 8;
			}
			// The 24 closing terms of a lane (6 harmonics x {re, im} x {main, difference window}) are summed over the group's eight
			// lanes through LDS, one harmonic at a time: every lane leaves its four terms of the harmonic in `red` (two planes of
			// 16-byte slots, slot = 8 sub + a rotation of the group so that neither the writes nor the reads below conflict), lane
			// (group, c) adds up term c of the eight lanes in the order of the halving butterfly this replaces --
			// ((x0 + x4) + (x2 + x6)) + ((x1 + x5) + (x3 + x7)), additions commute -- and the four sums wait in this pass's dead
			// start-phase entry (harmonics 0-3) or in registers (4, 5) for lane h to collect them.  ~110 instructions where the
			// butterfly's 28 exchanges of selects and DPP moves took ~300.  A wavefront's LDS operations execute in order: no fences.
			double *const res = reinterpret_cast<double *>(&stage[q][0]);  // [grp][h < 4][c]
			const int jc = sub_c & 3;
			const int wslot = sub_c * 8 + ((grp_c + 2 * (sub_c >> 1)) & 7);
			int rbase[4];
#pragma unroll
			for (int m = 0; m < 4; ++m) rbase[m] = (jc >> 1) * RF_RED_PLANE + (jc & 1) + 2 * (16 * m + ((grp_c + 2 * m) & 7));
			// the two closing twiddles of harmonic h (table indices are multiples of a power of two: shifts, not multiplications);
			// a harmonic's pair is requested one round ahead of its use
			const int tsl_c = __builtin_ctz((unsigned)tsh_c);
			auto twiddles = [&](int h, double2 &e1, double2 &e2) {
				const int i1 = (idx[h] * (sub_c + 8 * (Q - 1))) & (N_c - 1);
				e1 = a.tw[i1 << tsl_c];
				e2 = a.tw[((i1 + 8 * idx[h]) & (N_c - 1)) << tsl_c];
			};
			double2 e1n, e2n;
			twiddles(0, e1n, e2n);
			double r45[2] = {0.0, 0.0};
#pragma unroll
			for (int h = 0; h < 6; ++h) {
				const double2 e1 = e1n, e2 = e2n;
				if (h < 5) twiddles(h + 1, e1n, e2n);
				double o[4];
				o[0] = sa[2 * h] * e1.x - sb[2 * h] * e2.x;
				o[1] = sb[2 * h] * e2.y - sa[2 * h] * e1.y;
				o[2] = sa[2 * h + 1] * e1.x - sb[2 * h + 1] * e2.x;
				o[3] = sb[2 * h + 1] * e2.y - sa[2 * h + 1] * e1.y;
				*reinterpret_cast<double2 *>(&red[2 * wslot]) = make_double2(o[0], o[1]);
				*reinterpret_cast<double2 *>(&red[RF_RED_PLANE + 2 * wslot]) = make_double2(o[2], o[3]);
				double x[8];
#pragma unroll
				for (int s_ = 0; s_ < 8; ++s_) x[s_] = red[rbase[s_ >> 1] + 16 * (s_ & 1)];
				const double sum = ((x[0] + x[4]) + (x[2] + x[6])) + ((x[1] + x[5]) + (x[3] + x[7]));
				if (h < 4) {
					if (sub_c < 4) res[(grp_c * 4 + h) * 4 + jc] = sum;
				} else {
					r45[h - 4] = sum;
				}
			}
			if (sub_c < 4) {
				red[(grp_c * 2 + 0) * 4 + jc] = r45[0];
				red[(grp_c * 2 + 1) * 4 + jc] = r45[1];
			}
			const int h = sub_c;
			int myidx = 0;
#pragma unroll
			for (int q2 = 0; q2 < 6; ++q2) if (q2 == h) myidx = idx[q2];
			const double *const mine4 = h < 4 ? &res[(grp_c * 4 + h) * 4] : &red[(grp_c * 2 + ((h - 4) & 1)) * 4];
			const double2 m01 = *reinterpret_cast<const double2 *>(mine4), m23 = *reinterpret_cast<const double2 *>(mine4 + 2);
			const double mr = m01.x, mi = m01.y, dr = m23.x, di = m23.y;
			const double pw = mr * mr + mi * mi;
			const double ni = mr * di - mi * dr;
			const double inst = (pw == 0.0) ? 0.0 : ldexp((double)myidx * fs, -__builtin_ctz((unsigned)N_c)) + ni / pw * fs / 2.0 / kPi;  // (N is a power of two: scaling by its exponent is the quotient of :871)
			const double amp = sqrt(pw);
			if (sub_c < 6) stage[q][grp_c * 6 + sub_c] = make_double2(inst, amp);  // (every lane took its start phase from here long ago)
			double num, den, sc;
			finish(ln, inst, amp, it_f[t_c], it_nh[t_c], num, den, sc);
			if (sub_c == 0 && live_c) {
				const int at = it_pos[t_c];
				row_f[at] = num;
				row_s[at] = den;
				row_c[at] = sc;
			}
		}
		__syncthreads();
		// 4. candidates that share the key of one of this group's passes: its harmonics, their own frequency
		for (int d0 = 0; d0 < nd; d0 += 8) {
			const int k = d0 + grp;
			const int t = k < nd ? dup_of[k] : 0;
			const int r = (int)it_u[t] - c0 * 8;
			const bool mine = k < nd && r >= 0 && r < cn * 8;
			if (__ballot(mine) == 0ull) continue;
			double inst = 0.0, amp = 0.0;
			if (mine && sub < 6) {
				const double2 ia = stage[r >> 3][(r & 7) * 6 + sub];
				inst = ia.x;
				amp = ia.y;
			}
			const double fc = it_f[t];
			double num, den, sc;
			finish(lane, inst, amp, fc, it_nh[t], num, den, sc);
			if (sub == 0 && mine) {
				const int at = it_pos[t];
				row_f[at] = num;
				row_s[at] = den;
				row_c[at] = sc;
			}
		}
		__syncthreads();
	}
	// refined F0 and score of every position (reference :964-979), one lane each; the rows leave contiguously
	for (int k = lane; k < NC; k += 64) {
		const int nh = pos_nh[k];
		double rf = 0.0, rs = 0.0;
		if (nh > 0) {
			rf = row_f[k] / (row_s[k] + kSafeH);
			rs = 1.0 / (row_c[k] / nh + kSafeH);
			if (rf < a.p.f0_floor || rf > a.p.f0_ceil || rs < 2.5) { rf = 0.0; rs = 0.0; }  // reference :974-979
		}
		a.cand1[g * a.p.n_cand + k] = rf;
		a.score1[g * a.p.n_cand + k] = rs;
	}
}


// The packed refinement with the work of RQ_F neighbouring frames dealt out together (round 6, default; the kernel above is
// WC_HARVEST_REFINE=packed).  A pass of the kernel above takes as long as the longest of its eight windows and a frame's last pass is
// rarely full: counted on the headline signal a frame has 14.5 distinct keys with windows of 40 .. 340 samples -- 21.9 trips of the
// sample loop per frame in 2.25 passes where eight lanes per key, perfectly packed, would need 13.8.  Here a workgroup of RQ_F
// wavefronts takes RQ_F consecutive frames: each wavefront gathers and de-duplicates its own frame as above, the distinct keys of all
// the frames are ranked by window length, and the passes are cut from that one list -- eight neighbours of it have windows within a
// few samples of each other, and only the group's last pass is partly empty: 15.5 trips and 1.91 passes per frame at four frames.
// Behind the ranking the wavefronts no longer wait for each other: each takes the next pass of the list when it is through with
// its own (longest first, so they finish together), scores the key's candidate and every candidate of the same frame that shares
// the key -- lane m of the group takes member m, summing the six harmonics in the reference's order -- and stores refined F0 and
// score straight into the frame's rows (the positions without a candidate were zeroed by the frame's own wavefront).
// The arithmetic of a key is that of the kernel above instruction for instruction (eight lanes, samples n = sub + 8 q, the same
// closing sums): same bits, whatever group of whatever wavefront takes it.
#ifndef WC_RQ_F
#define WC_RQ_F 4
#endif
#ifndef WC_RQ_PC
#define WC_RQ_PC 8   // passes whose start phases are staged together: 8 KB, one stretch for up to 64 keys
#endif
// NP: candidate positions per frame the LDS is sized for: 112 (15 slots: the reference's default floor of 71 Hz; four workgroups per
// CU) or 128 (18 slots: the 40 Hz floor of the reference's demo; three per CU); wider rows take the kernel above
template <bool TABLE, int F, int NP = 112>
__global__ __launch_bounds__(64 * F, WC_REFINE_WAVES) void hv_refine_group_kernel(RefArgs a) {
	static_assert(NP <= 128, "positions are gathered in two rounds of 64 and travel as seven bits");
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int grp = lane >> 3, sub = lane & 7;
	// a workgroup per F consecutive frames of ONE utterance (blockIdx.y), the groups of an utterance XCD by XCD (see above)
	const HvUtt u = a.utts[blockIdx.y];
	const int n_grp = (u.L1 + F - 1) / F, per = (n_grp + 7) / 8;
	if ((int)(blockIdx.x >> 3) >= per) return;
	const int G = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
	if (G >= n_grp) return;
	const int i0 = G * F, i = i0 + wv;
	const bool have = i < u.L1;
	const long long g0 = u.l1_off + i0, g = g0 + wv;
	const double fs = a.p.fs_d;
	const int S = a.p.S;
	const int NC = 7 * S;
	const double *__restrict__ y = a.y + u.y_off;
	const double *__restrict__ crow = a.cand0 + g * S;
	__shared__ double it_f[F][NP];
	__shared__ unsigned long long key[F][NP];
	__shared__ __attribute__((aligned(16))) double red[F][RF_RED];  // (the de-duplication's hash table lives here before the first pass)
	__shared__ double2 stage[F][64];  // a wavefront's closing sums on their way to the harmonics' lanes
	__shared__ double2 it_ph[F][NP];  // window phase of a distinct key at its sample 0
	__shared__ int it_basic[F][NP];
	__shared__ unsigned char it_pos[F][NP], un_src[F][NP], it_nh[F][NP], un_at[F][NP], dup_next[F][NP];
	__shared__ int dup_head[F][NP];  // the candidates of a frame that share the key of candidate t: dup_head[t], dup_next[that], ... (>= NP: none)
	__shared__ int hist[F][128];   // distinct keys per wavefront and window-length class (eight samples of half length to a class, longest first)
	__shared__ int place[F][128];  // where a wavefront's keys of a class start in the group's list
	__shared__ unsigned short w_item[F * NP];
	__shared__ int fr_n[F];
	__shared__ int next_pass;
	const unsigned long long below = (1ull << lane) - 1ull;
#if WC_RQ_PROF
	long long acc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
	long long last_ = clock64();
#endif
	hist[wv][lane] = 0;
	hist[wv][64 + lane] = 0;

	// 1. live candidates of the overlap (reference :987-1000) in the order of the frame's row: position p = S block + j (the order
	//    decides nothing but which of several candidates with one key stands for it).  Both rounds' rows are requested up front.
	int n = 0;
	{
		const double pos = i * 1 / 1000.0;
		double fv[2] = {0.0, 0.0};
		const float r_S = 1.0f / (float)S;
#pragma unroll
		for (int rd = 0; rd < 2; ++rd) {
			const int p_ = rd * 64 + lane;
			const int blk = (int)((p_ + 0.5f) * r_S), j = p_ - blk * S;  // (p_ / S: exact for these small integers)
			const int src = (blk == 0) ? i : (blk <= 3 ? i - blk : i + (blk - 3));
			if (have && p_ < NC && src >= 0 && src < u.L1) fv[rd] = crow[(src - i) * S + j];
		}
#pragma unroll
		for (int rd = 0; rd < 2; ++rd) {
			const int p_ = rd * 64 + lane;
			if (!have || rd * 64 >= NC) break;
			const double f = fv[rd];
			const bool live = f > 0.0;
			const unsigned long long m = __ballot(live);
			if (p_ < NC && !live) {  // an empty position
				a.cand1[g * a.p.n_cand + p_] = 0.0;
				a.score1[g * a.p.n_cand + p_] = 0.0;
			}
			if (live) {
				const int at = n + __popcll(m & below);
				const int hw = min((int)(1.5 * fs / f + 1.0), RF_MAXHW);
				const int N = 1 << (2 + (31 - __clz(hw * 2 + 1)));
				const double bin_unit = f * N / fs;
				const int b0 = mround(bin_unit);
				unsigned long long kk = (unsigned long long)hw | ((unsigned long long)(unsigned)b0 << 32);  // (the key of the kernel above)
#pragma unroll
				for (int h = 1; h < 6; ++h) {
					const int dlt = mround(bin_unit * (h + 1)) - (h + 1) * b0 + 8;
					kk |= (unsigned long long)(dlt & 15) << (11 + 4 * (h - 1));
				}
				it_f[wv][at] = f;
				it_pos[wv][at] = (unsigned char)p_;
				key[wv][at] = kk;
				it_basic[wv][at] = mround((pos + (-hw) / fs) * fs + 0.001);
				it_nh[wv][at] = (unsigned char)min((int)(fs / 2.0 / f), 6);
				{  // ties: see the kernel above
					const double tol = 2e-13;
					const double v = 1.5 * fs / f + 1.0, v2 = fs / 2.0 / f;
					bool tie = fabs(v - rint(v)) < tol * v || (v2 < 7.0 && fabs(v2 - rint(v2)) < tol * v2);
#pragma unroll
					for (int h = 0; h < 6; ++h) {
						const double wv_ = bin_unit * (h + 1);
						tie = tie || fabs(wv_ - floor(wv_) - 0.5) < tol * wv_;
					}
					if (tie) { a.flags[1] = 1; a.flags[2 + blockIdx.y] = 1; }
				}
			}
			n += __popcll(m);
		}
	}
	RQ_T(0);
	// 2. the first candidate of every key (the hash table of the kernel above; a wavefront's LDS operations execute in order), and
	//    the distinct keys counted by window-length class
	int nu = 0;
	if (n > 0) {
		unsigned int *const tab = reinterpret_cast<unsigned int *>(&red[wv][0]);
		auto slot_of = [](unsigned long long kk) { return (((unsigned)kk * 2654435761u) ^ ((unsigned)(kk >> 32) * 40503u * 65537u)) >> 24; };
		for (int k = lane; k < 256; k += 64) tab[k] = 0xFFFFFFFFu;
		rq_fence();
		for (int t = lane; t < n; t += 64) atomicMin(&tab[slot_of(key[wv][t])], (unsigned)t);
		rq_fence();
		for (int t0 = 0; t0 < n; t0 += 64) {
			const int t = t0 + lane;
			int rep = t;
			unsigned long long mine = 0;
			if (t < n) {
				mine = key[wv][t];
				const int r = (int)tab[slot_of(mine)];
				if (r != t && key[wv][r] == mine) rep = r;
			}
			const bool uniq = t < n && rep == t, dup = t < n && rep != t;
			const unsigned long long mu = __ballot(uniq);
			if (uniq) {
				const int r = nu + __popcll(mu & below);
				un_src[wv][r] = (unsigned char)t;
				un_at[wv][r] = (unsigned char)atomicAdd(&hist[wv][127 - (int)((mine & 2047ull) >> 3)], 1);  // (its place among the wavefront's keys of the class)
			}
			if (uniq) dup_head[wv][t] = 255;
			if (dup) dup_next[wv][t] = (unsigned char)atomicExch(&dup_head[wv][rep], t);  // (rep < t: its head has been set, in this round by the line above)
			nu += __popcll(mu);
		}
	}
	// the window phase of every distinct key at sample 0: a lane per key here, where the kernels above spend a lane per (key, first
	// sample) and pass
	for (int r = lane; r < nu; r += 64) {
		const int t = un_src[wv][r];
		it_ph[wv][t] = rf_phase0((int)(key[wv][t] & 2047ull), it_basic[wv][t], i * 1 / 1000.0, fs);
	}
	if (lane == 0) fr_n[wv] = nu;
	if (threadIdx.x == 0) next_pass = 0;
	RQ_T(1);
	__syncthreads();
	RQ_T(2);
	// 3. one list of the group's distinct keys, longest windows first: a key's place is the number of keys in the classes before
	//    its own, plus those of its class in the wavefronts before this one, plus its place among this wavefront's
	int M = 0;
#pragma unroll
	for (int w = 0; w < F; ++w) M += fr_n[w];
	if (M == 0) { RQ_FLUSH(); return; }
	{
		int tot2[2], mine2[2];  // classes 2 lane, 2 lane + 1: keys in all wavefronts / in the wavefronts before this one
#pragma unroll
		for (int c = 0; c < 2; ++c) {
			tot2[c] = 0; mine2[c] = 0;
#pragma unroll
			for (int w = 0; w < F; ++w) {
				const int hv_ = hist[w][2 * lane + c];
				tot2[c] += hv_;
				if (w < wv) mine2[c] += hv_;
			}
		}
		int inc = tot2[0] + tot2[1];
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const int t = __shfl_up(inc, o, 64);
			if (lane >= o) inc += t;
		}
		const int ex = inc - tot2[0] - tot2[1];
		place[wv][2 * lane] = ex + mine2[0];
		place[wv][2 * lane + 1] = ex + tot2[0] + mine2[1];
		rq_fence();
	}
	for (int r = lane; r < nu; r += 64) {
		const int t = un_src[wv][r];
		const int rk = place[wv][127 - (int)((key[wv][t] & 2047ull) >> 3)] + un_at[wv][r];
		w_item[rk] = (unsigned short)((wv << 7) | t);
	}
	RQ_T(3);

	auto finish = [&](int ln, double inst, double amp, double fc, int nh, double &num, double &den, double &sc) {  // (see the kernel above)
		const int h = ln & 7;
		const bool on = h < nh;
		const double e_num = on ? amp * inst : 0.0, e_den = on ? amp * (h + 1.0) : 0.0, e_sc = on ? fabs((inst / (h + 1.0) - fc) / fc) : 0.0;
		num = 0.0 + e_num; den = 0.0 + e_den; sc = 0.0 + e_sc;
		num += row_shl_d<1>(e_num); den += row_shl_d<1>(e_den); sc += row_shl_d<1>(e_sc);
		num += row_shl_d<2>(e_num); den += row_shl_d<2>(e_den); sc += row_shl_d<2>(e_sc);
		num += row_shl_d<3>(e_num); den += row_shl_d<3>(e_den); sc += row_shl_d<3>(e_sc);
		num += row_shl_d<4>(e_num); den += row_shl_d<4>(e_den); sc += row_shl_d<4>(e_sc);
		num += row_shl_d<5>(e_num); den += row_shl_d<5>(e_den); sc += row_shl_d<5>(e_sc);
	};
	// refined F0 and score of candidate t of frame w from its three sums (reference :964-979), stored by the lane that holds them
	auto store = [&](int w, int t, double num, double den, double sc) {
		const int nh = it_nh[w][t];
		double rf = num / (den + kSafeH);
		double rs = 1.0 / (sc / nh + kSafeH);
		if (rf < a.p.f0_floor || rf > a.p.f0_ceil || rs < 2.5) { rf = 0.0; rs = 0.0; }
		const long long at = (g0 + w) * a.p.n_cand + it_pos[w][t];
		a.cand1[at] = rf;
		a.score1[at] = rs;
	};

	const int npass = (M + 7) >> 3;
	constexpr int c0 = 0;
	const int q_own = wv;  // (this wavefront's entry of `stage`)
	__syncthreads();  // (the list is complete)
	RQ_T(5);
	{
		// a wavefront takes the next pass of the list when it is through with its own
		for (;;) {
			int q = 0;
			if (lane == 0) q = atomicAdd(&next_pass, 1);
			q = __builtin_amdgcn_readfirstlane(q);
			RQ_T(6);
			if (q >= npass) break;
			const int r = (c0 + q) * 8 + grp;
			const bool live = r < M;
			const int item = w_item[live ? r : 0];
			const int ws = item >> 7, t_own = item & 127;
			const unsigned long long kk = key[ws][t_own];
			const int hw = (int)(kk & 2047ull);
			const int bt = live ? 2 * hw + 1 : 0;
			const double wlt = (2.0 * hw + 1.0) / fs;
			const int fft_index = 2 + (31 - __clz(hw * 2 + 1));
			const int N = 1 << fft_index;
			const int basic = it_basic[ws][t_own];
			const double pos = (i0 + ws) * 1 / 1000.0;
			auto bin_of = [&](int h) -> int {
				const int b0 = (int)(kk >> 32);
				return h == 0 ? b0 : (h + 1) * b0 + (int)(((unsigned)kk >> (11 + 4 * (h - 1))) & 15u) - 8;
			};
			const double2 r1 = a.rot[2 * hw], r8 = a.rot[2 * hw + 1];
			const double2 ph = rf_turn(it_ph[ws][t_own], a.rot8[hw * 8 + sub]);
			double wc = ph.x, ws_ = ph.y;
			const double k1 = 0.5 * r1.y, k2 = 0.16 * (2.0 * r1.y * r1.x);
			double c2[6];
#pragma unroll
			for (int h = 0; h < 6; ++h) c2[h] = 2.0 * a.tw[((bin_of(h) * 8) & (N - 1)) << (kTwiddleLog2 - fft_index)].x;
			double sa[12], sb[12];
#pragma unroll
			for (int k = 0; k < 12; ++k) { sa[k] = 0.0; sb[k] = 0.0; }
			auto table_window = [&](int n_) -> double {  // reference :779-787, operation by operation
				const double two_pi = 2.0 * kPi;
				const double tmp = (basic + n_ - 1.0) / fs - pos;
				const double tmp2 = two_pi * (tmp / wlt + 1);
				const double dindex = fmod(tmp2, two_pi) / two_pi * 8000;
				const double dindex2 = fmod(dindex * 2, 8000.0);
				return 0.42 + 0.5 * a.cos_table[(int)round(dindex)] + 0.08 * a.cos_table[(int)round(dindex2)];
			};
			auto sample = [&](int n_, double yl, double &xm, double &xd) {
				const double yv = (n_ < bt) ? yl : 0.0;
				if (TABLE) {
					double m = 0.0, d = 0.0;
					if (n_ < bt) {
						m = table_window(n_);
						if (n_ == 0) d = -table_window(1) / 2.0;
						else if (n_ == bt - 1) d = table_window(bt - 2) / 2.0;
						else d = -(table_window(n_ + 1) - table_window(n_ - 1)) / 2.0;
					}
					xm = m * yv;
					xd = d * yv;
					return;
				}
				const double m = fma(wc, fma(0.16, wc, 0.5), 0.34);
				const bool first = n_ == 0, last = n_ == bt - 1;
				const double cnb = fma(wc, r1.x, first ? -(ws_ * r1.y) : ws_ * r1.y);
				const double mnb = fma(cnb, fma(0.16, cnb, 0.5), 0.34);
				double d = ws_ * fma(k2, wc, k1);
				d = first ? -mnb / 2.0 : (last ? mnb / 2.0 : d);
				xm = m * yv;
				xd = d * yv;
				const double nc_ = fma(wc, r8.x, -(ws_ * r8.y));
				ws_ = fma(ws_, r8.x, wc * r8.y);
				wc = nc_;
			};
			auto interior = [&](int n_, double yv, double &xm, double &xd) {
				const double m = fma(wc, fma(0.16, wc, 0.5), 0.34);
				const double d = ws_ * fma(k2, wc, k1);
				xm = m * yv;
				xd = d * yv;
				const double nc_ = fma(wc, r8.x, -(ws_ * r8.y));
				ws_ = fma(ws_, r8.x, wc * r8.y);
				wc = nc_;
			};
			RQ_T(7);
			auto sample_at = [&](int n_) { return y[clampi(basic + n_ - 1, 0, u.y_len - 1)]; };
			double ya = sample_at(sub), yb = sample_at(sub + 8);
			int Q = 0;
			for (int n_ = sub; n_ < bt; n_ += 16) {
				const double nya = sample_at(n_ + 16), nyb = sample_at(n_ + 24);
				const bool ends = TABLE || n_ == sub || n_ + 9 >= bt;
				const bool any_end = __ballot(ends) != 0ull;
				double xm, xd;
				if (any_end) sample(n_, ya, xm, xd);
				else interior(n_, ya, xm, xd);
#pragma unroll
				for (int h = 0; h < 6; ++h) {
					sb[2 * h] = fma(c2[h], sa[2 * h], xm) - sb[2 * h];
					sb[2 * h + 1] = fma(c2[h], sa[2 * h + 1], xd) - sb[2 * h + 1];
				}
				if (any_end) sample(n_ + 8, yb, xm, xd);
				else interior(n_ + 8, yb, xm, xd);
#pragma unroll
				for (int h = 0; h < 6; ++h) {
					sa[2 * h] = fma(c2[h], sb[2 * h], xm) - sa[2 * h];
					sa[2 * h + 1] = fma(c2[h], sb[2 * h + 1], xd) - sa[2 * h + 1];
				}
				Q += 2;
				ya = nya;
				yb = nyb;
			}
			RQ_T(8);
			// (everything the closing needs is looked up again from an opaque copy of the lane index: see the kernel above)
			int ln = lane;
			asm volatile("" : "+v"(ln));
			const int sub_c = ln & 7, grp_c = ln >> 3;
			const int r_c = (c0 + q) * 8 + grp_c;
			const bool live_c = r_c < M;
			const int item_c = w_item[live_c ? r_c : 0];
			const int ws_c = item_c >> 7, t_c = item_c & 127;
			const unsigned long long kk_c = key[ws_c][t_c];
			const int N_c = 1 << (2 + (31 - __clz((int)(kk_c & 2047ull) * 2 + 1)));
			const int tsh_c = kTwiddleN / N_c;
			int idx[6];
#pragma unroll
			for (int h = 0; h < 6; ++h) {
				const int b0 = (int)(kk_c >> 32);
				idx[h] = h == 0 ? b0 : (h + 1) * b0 + (int)(((unsigned)kk_c >> (11 + 4 * (h - 1))) & 15u) - 8;
			}
			double *const res = reinterpret_cast<double *>(&stage[q_own][0]);  // [grp][h < 4][c]
			double *const redw = &red[wv][0];
			const int jc = sub_c & 3;
			const int wslot = sub_c * 8 + ((grp_c + 2 * (sub_c >> 1)) & 7);
			int rbase[4];
#pragma unroll
			for (int m = 0; m < 4; ++m) rbase[m] = (jc >> 1) * RF_RED_PLANE + (jc & 1) + 2 * (16 * m + ((grp_c + 2 * m) & 7));
			const int tsl_c = __builtin_ctz((unsigned)tsh_c);
			// the closing twiddles of the six harmonics, all requested before the first is used
			double2 e1a[6], e2a[6];
#pragma unroll
			for (int h = 0; h < 6; ++h) {
				const int i1 = (idx[h] * (sub_c + 8 * (Q - 1))) & (N_c - 1);
				e1a[h] = a.tw[i1 << tsl_c];
				e2a[h] = a.tw[((i1 + 8 * idx[h]) & (N_c - 1)) << tsl_c];
			}
			double r45[2] = {0.0, 0.0};
#pragma unroll
			for (int h = 0; h < 6; ++h) {
				const double2 e1 = e1a[h], e2 = e2a[h];
				double o[4];
				o[0] = sa[2 * h] * e1.x - sb[2 * h] * e2.x;
				o[1] = sb[2 * h] * e2.y - sa[2 * h] * e1.y;
				o[2] = sa[2 * h + 1] * e1.x - sb[2 * h + 1] * e2.x;
				o[3] = sb[2 * h + 1] * e2.y - sa[2 * h + 1] * e1.y;
				*reinterpret_cast<double2 *>(&redw[2 * wslot]) = make_double2(o[0], o[1]);
				*reinterpret_cast<double2 *>(&redw[RF_RED_PLANE + 2 * wslot]) = make_double2(o[2], o[3]);
				double x[8];
#pragma unroll
				for (int s_ = 0; s_ < 8; ++s_) x[s_] = redw[rbase[s_ >> 1] + 16 * (s_ & 1)];
				const double sum = ((x[0] + x[4]) + (x[2] + x[6])) + ((x[1] + x[5]) + (x[3] + x[7]));
				if (h < 4) {
					if (sub_c < 4) res[(grp_c * 4 + h) * 4 + jc] = sum;
				} else {
					r45[h - 4] = sum;
				}
			}
			if (sub_c < 4) {
				redw[(grp_c * 2 + 0) * 4 + jc] = r45[0];
				redw[(grp_c * 2 + 1) * 4 + jc] = r45[1];
			}
			const int h = sub_c;
			int myidx = 0;
#pragma unroll
			for (int q2 = 0; q2 < 6; ++q2) if (q2 == h) myidx = idx[q2];
			const double *const mine4 = h < 4 ? &res[(grp_c * 4 + h) * 4] : &redw[(grp_c * 2 + ((h - 4) & 1)) * 4];
			const double2 m01 = *reinterpret_cast<const double2 *>(mine4), m23 = *reinterpret_cast<const double2 *>(mine4 + 2);
			const double mr = m01.x, mi = m01.y, dr = m23.x, di = m23.y;
			const double pw = mr * mr + mi * mi;
			const double ni = mr * di - mi * dr;
			const double inst = (pw == 0.0) ? 0.0 : ldexp((double)myidx * fs, -__builtin_ctz((unsigned)N_c)) + ni / pw * fs / 2.0 / kPi;
			const double amp = sqrt(pw);
			RQ_T(9);
			// the key's own candidate, then -- a round for the first of every group, a round for the second, ... -- the candidates of
			// the same frame that share the key: the same harmonics, their own frequency (fixF0, reference :880-893)
			{
				double num, den, sc;
				finish(ln, inst, amp, it_f[ws_c][t_c], it_nh[ws_c][t_c], num, den, sc);
				if (sub_c == 0 && live_c) store(ws_c, t_c, num, den, sc);
			}
			int t_m = live_c ? dup_head[ws_c][t_c] : 255;
			while (__ballot(t_m < NP) != 0ull) {
				const int t_s = t_m < NP ? t_m : t_c;
				double num, den, sc;
				finish(ln, inst, amp, it_f[ws_c][t_s], it_nh[ws_c][t_s], num, den, sc);
				if (sub_c == 0 && t_m < NP) store(ws_c, t_s, num, den, sc);
				t_m = t_m < NP ? dup_next[ws_c][t_s] : 255;
			}
			RQ_T(10);
		}
	}
	RQ_FLUSH();
}

// reference :708-744.  One workgroup per UNR_F consecutive frames of an utterance: the rows of the UNR_F + 2 frames involved are
// compacted into LDS once (non-zero candidates in slot order; a zero candidate yields the error 1.0, which selectBestF0's allowed
// range of 1.0 already is), then every candidate scans the lists of the two neighbouring frames.
// min_k fl(|ref - x_k| / ref) = fl((min_k |ref - x_k|) / ref) -- rounding is monotone -- so the scan takes differences only and
// divides once.  searchF0Base (:254-272) on the surviving candidates is a reduction (highest score, first slot on ties).
// A frame's own candidates are compacted as well (most of its 7 S positions are empty) and half a wavefront takes a frame -- eight
// frames at once per workgroup; the rows travel through LDS once: read coalesced, candidates struck out in place, written back
// coalesced.  LDS is sized by the row width at launch: (UNR_F + 2) lists + 2 UNR_F rows of nc doubles, 22 KB at the default nc = 105.
// (Round 1 ran one workgroup per frame: three row reads, four barriers and two LDS-atomic compactions per frame; until late in round 2
// a wavefront took a frame with its lanes over all 7 S positions, a fifth of them live: 2.46 -> 2.13 ms for the tail of 64 utterances.)
constexpr int UNR_F = 8;

__global__ __launch_bounds__(256) void hv_unreliable_kernel(const HvUtt *__restrict__ utts, const double *__restrict__ c1,
									 const double *__restrict__ s1, double *__restrict__ c2, double *__restrict__ s2,
									 double *__restrict__ base, int nc) {
	extern __shared__ double unr_lds[];
	double *const lst = unr_lds;                        // [UNR_F + 2][nc] non-zero candidates of frames first - 1 .. first + UNR_F
	double *const row_v = lst + (UNR_F + 2) * nc;       // [UNR_F][nc] the frames' own rows
	double *const row_s = row_v + UNR_F * nc;
	int *const cnt = reinterpret_cast<int *>(row_s + UNR_F * nc);  // [UNR_F + 2] list lengths
	int *const own_n = cnt + UNR_F + 2;                 // [UNR_F] live positions of a frame's own row
	unsigned char *const own_j = reinterpret_cast<unsigned char *>(own_n + UNR_F);  // [UNR_F][nc] ... and where they are
	const HvUtt u = utts[blockIdx.y];
	const int first = blockIdx.x * UNR_F;
	if (first >= u.L1) return;
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const unsigned long long below = (1ull << lane) - 1ull;
	for (int r = wv; r < UNR_F + 2; r += 4) {
		const int i = first - 1 + r;
		const bool exists = i >= 0 && i < u.L1;
		// the reference's comparison copy holds frames 1 .. L-2 only (:714-715); its rows 0 and L-1 are never written
		// (uninitialised there; zero here, as with a zero-filling allocator under the reference)
		const bool held = i >= 1 && i <= u.L1 - 2;
		const bool own = r >= 1 && r <= UNR_F;
		int n = 0;
		for (int j0 = 0; j0 < nc; j0 += 64) {
			const int j = j0 + lane;
			double v = 0.0, sc = 0.0;
			if (exists && j < nc) {
				v = c1[(u.l1_off + i) * nc + j];
				if (own) sc = s1[(u.l1_off + i) * nc + j];
			}
			const unsigned long long m = __ballot(v != 0.0);
			const int at = n + __popcll(m & below);
			if (v != 0.0) {
				if (held) lst[r * nc + at] = v;
				if (own) own_j[(r - 1) * nc + at] = (unsigned char)j;
			}
			if (own && j < nc) { row_v[(r - 1) * nc + j] = v; row_s[(r - 1) * nc + j] = sc; }
			n += __popcll(m);
		}
		if (lane == 0) {
			cnt[r] = held ? n : 0;
			if (own) own_n[r - 1] = n;
		}
	}
	__syncthreads();
	const int k = threadIdx.x >> 5, l32 = threadIdx.x & 31;  // half a wavefront per frame
	const int i = first + k;
	const bool interior = i >= 1 && i < u.L1 - 1;
	if (interior) {
		const int n_prev = cnt[k], n_next = cnt[k + 2];
		const double *__restrict__ prv = lst + k * nc, *__restrict__ nxt = lst + (k + 2) * nc;
		for (int q0 = l32; q0 < own_n[k]; q0 += 32) {
			const int j = own_j[k * nc + q0];
			const double ref = row_v[k * nc + j];
			double dmin = ref;  // |ref - 0| / ref = 1.0, the allowed range
			for (int q = 0; q < n_next; ++q) dmin = fmin(dmin, fabs(ref - nxt[q]));
			for (int q = 0; q < n_prev; ++q) dmin = fmin(dmin, fabs(ref - prv[q]));
			if (fmin(1.0, dmin / ref) > 0.05) { row_v[k * nc + j] = 0.0; row_s[k * nc + j] = 0.0; }
		}
	}
	__syncthreads();
	if (i < u.L1) {
		const long long g = u.l1_off + i;
		double top_sc = 0.0, top_ref = 0.0;
		int top_slot = 0x7fffffff;
		for (int j = l32; j < nc; j += 32) {
			const double ref = row_v[k * nc + j], sc = row_s[k * nc + j];
			c2[g * nc + j] = ref;
			s2[g * nc + j] = sc;
			if (sc > top_sc) { top_sc = sc; top_ref = ref; top_slot = j; }  // (a lane's slots come in ascending order)
		}
#pragma unroll
		for (int o = 16; o > 0; o >>= 1) {  // searchF0Base (:254-272): highest score, first slot on ties, within the half wavefront
			const double osc = __shfl_xor(top_sc, o, 64), oref = __shfl_xor(top_ref, o, 64);
			const int oslot = __shfl_xor(top_slot, o, 64);
			if (osc > top_sc || (osc == top_sc && oslot < top_slot)) { top_sc = osc; top_ref = oref; top_slot = oslot; }
		}
		if (l32 == 0) base[g] = top_sc > 0.0 ? top_ref : 0.0;
	}
}

// ------------------------------------------------------------------------------------------------
// contour fixing: one wavefront per utterance
// ------------------------------------------------------------------------------------------------
struct CtrArgs {
	const HvUtt *utts;
	const double *cand, *score;  // after removeUnreliableCandidates
	double *base, *s1, *s2, *s3, *fixed;  // [total L1 frames] work contours
	int *sec;         // [utt][2 * max_sec] section boundaries (st, ed)
	double *chan;     // channel storage
	long long chan_stride;  // doubles per utterance
	int max_sec;
	int nc;
	int *ibuf;        // [utt][4 * max_sec] position bookkeeping of fixStep3
	int *nsec;        // [utt] voiced sections after fixStep2, handed from phase to phase
};

__device__ __forceinline__ void wave_sync() {
	__syncthreads();  // 64-thread workgroup: waitcnt + barrier
}

// argmin_k |ref - c[k]| / ref over the lanes with the reference's tie rule (the last smallest error that
// does not exceed `allowed` wins), reference :347-365
// (the candidates of a row arrive in registers -- lane l holds c[l] and c[l + 64] -- so that callers can request the
// next row while this one is being searched, and the winner's value travels with the reduction instead of being
// re-read from memory)
__device__ __forceinline__ double hv_select_best_regs(double ref, double c0, double c1, int nc, double allowed, int lane) {
	double best_err = allowed, best_v = 0.0;
	int best_k = -1;
	if (lane < nc) {
		double t = fabs(ref - c0) / ref;
		if (!(t > best_err)) { best_err = t; best_k = lane; best_v = c0; }
	}
	if (lane + 64 < nc) {
		double t = fabs(ref - c1) / ref;
		if (!(t > best_err)) { best_err = t; best_k = lane + 64; best_v = c1; }
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		double oe = __shfl_xor(best_err, o, 64);
		double ov = __shfl_xor(best_v, o, 64);
		int ok = __shfl_xor(best_k, o, 64);
		if (ok >= 0 && (best_k < 0 || oe < best_err || (oe == best_err && ok > best_k))) { best_err = oe; best_k = ok; best_v = ov; }
	}
	return best_k >= 0 ? best_v : 0.0;
}
__device__ __forceinline__ double hv_select_best(double ref, const double *__restrict__ c, int nc, double allowed, int lane) {
	// nc <= 7 * MAX_SLOTS = 224 in general; two registers per lane cover nc <= 128, longer rows take the loop
	if (nc <= 128) {
		const double c0 = lane < nc ? c[lane] : 0.0, c1 = lane + 64 < nc ? c[lane + 64] : 0.0;
		return hv_select_best_regs(ref, c0, c1, nc, allowed, lane);
	}
	double best_err = allowed, best_v = 0.0;
	int best_k = -1;
	for (int k = lane; k < nc; k += 64) {
		const double cv = c[k];
		double t = fabs(ref - cv) / ref;
		if (!(t > best_err)) { best_err = t; best_k = k; best_v = cv; }
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		double oe = __shfl_xor(best_err, o, 64);
		double ov = __shfl_xor(best_v, o, 64);
		int ok = __shfl_xor(best_k, o, 64);
		if (ok >= 0 && (best_k < 0 || oe < best_err || (oe == best_err && ok > best_k))) { best_err = oe; best_k = ok; best_v = ov; }
	}
	return best_k >= 0 ? best_v : 0.0;
}
// reference :463-470
// (for two contour values of the same frame at once -- mergeF0Sub asks for both -- with the row requested eight candidates at
// a time: a lane walks its own frame's row, so every dependent load is a round trip of its own)
__device__ __forceinline__ void hv_search_score2(double fa, double fb, const double *__restrict__ c, const double *__restrict__ s, int nc,
												 double &score_a, double &score_b) {
	double sa = 0.0, sb = 0.0;
	for (int k0 = 0; k0 < nc; k0 += 8) {
		double cv[8], sv[8];
#pragma unroll
		for (int e = 0; e < 8; ++e) {
			const int k = min(k0 + e, nc - 1);
			cv[e] = c[k];
			sv[e] = s[k];
		}
#pragma unroll
		for (int e = 0; e < 8; ++e) {
			if (k0 + e < nc) {
				if (fa == cv[e] && sa < sv[e]) sa = sv[e];
				if (fb == cv[e] && sb < sv[e]) sb = sv[e];
			}
		}
	}
	score_a = sa;
	score_b = sb;
}

// init + p[0] + p[1] + ... + p[n-1] added strictly in that order (the reference's running sums), with the
// loads spread over the lanes: 64 values per step, then a 64-step dependent chain on broadcast values.
__device__ __forceinline__ double hv_readlane(double v, int k) {  // (k a constant: two v_readlane_b32, the sum then adds a scalar pair)
	return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), k), __builtin_amdgcn_readlane(__double2loint(v), k));
}
__device__ __forceinline__ double hv_ordered_sum(const double *__restrict__ p, int n, double init, int lane) {
	double run = init;
	double nxt = (lane < n) ? p[lane] : 0.0;
	for (int base = 0; base < n; base += 64) {
		const double v = nxt;  // + 0.0 past the end is exact
		const int i = base + 64 + lane;
		nxt = (i < n) ? p[i] : 0.0;  // (the next 64 values are in flight while these are added)
#pragma unroll
		for (int k = 0; k < 64; ++k) run = run + hv_readlane(v, k);
	}
	return run;
}

// boundaries of the voiced sections of f0[0..n) (reference :296-314): returns the number of sections;
// sec[2k] = first frame, sec[2k+1] = last frame.  Executed by all 64 lanes, ordered by ballot.
__device__ int hv_sections(const double *__restrict__ f0, int n, int *__restrict__ sec, int max_sec, int lane) {
	int nb = 0;  // boundaries so far (wave-uniform)
	constexpr int B = 4;  // batches of 64 frames whose loads are requested together (a single wavefront: latency is all there is)
	for (int base = 1; base < n; base += 64 * B) {
		double c[B], p[B];
#pragma unroll
		for (int b = 0; b < B; ++b) {
			const int i = base + 64 * b + lane;
			c[b] = i < n ? f0[i] : 0.0;
			p[b] = i < n ? f0[i - 1] : 0.0;
		}
#pragma unroll
		for (int b = 0; b < B; ++b) {
			const int i = base + 64 * b + lane;
			int cur = 0, prv = 0;
			if (i < n) {
				cur = (i >= 1 && i < n - 1 && c[b] > 0) ? 1 : 0;
				prv = (i - 1 >= 1 && i - 1 < n - 1 && p[b] > 0) ? 1 : 0;
			}
			const bool ch = (i < n) && (cur != prv);
			const unsigned long long m = __ballot(ch);
			if (ch) {
				int k = nb + __popcll(m & ((1ull << lane) - 1ull));
				if (k < 2 * max_sec) sec[k] = i - (k & 1);
			}
			nb += __popcll(m);
		}
	}
	return nb / 2;
}

// Three launches: PHASE 0 = fixStep1, fixStep2 and the channels of fixStep3 (one wavefront per utterance); PHASE 1 = the
// extension walks, which are independent per voiced section and direction (they read the section's original end, write
// outside it and update their own boundary), one wavefront each; PHASE 2 = extendSub, mergeF0, fixStep4 (one wavefront per
// utterance).  The walks were 70 % of the former single kernel.
template <int PHASE>
__global__ __launch_bounds__(64) void hv_contour_kernel(CtrArgs a) {
	const int ui = PHASE == 1 ? blockIdx.y : blockIdx.x;
	const HvUtt u = a.utts[ui];
	const int lane = threadIdx.x;
	const int L = u.L1, nc = a.nc;
	const double *__restrict__ cand = a.cand + u.l1_off * nc;
	const double *__restrict__ score = a.score + u.l1_off * nc;
	double *__restrict__ base = a.base + u.l1_off;
	double *__restrict__ s1 = a.s1 + u.l1_off;
	double *__restrict__ s2 = a.s2 + u.l1_off;
	double *__restrict__ s3 = a.s3 + u.l1_off;
	double *__restrict__ s4 = a.fixed + u.l1_off;
	int *__restrict__ sec = a.sec + (long long)ui * 2 * a.max_sec;
	double *__restrict__ chan = a.chan + (long long)ui * a.chan_stride;
	// channel k keeps frames [clo_k, clo_k + clen_k) around its section (extension moves at most 101 frames);
	// the reference's full-length rows are zero outside that window
	int *__restrict__ clo = reinterpret_cast<int *>(chan);
	int *__restrict__ clen = clo + a.max_sec;
	int *__restrict__ coff = clen + a.max_sec;
	int *__restrict__ perm = a.ibuf + (long long)ui * 4 * a.max_sec;  // position -> channel
	int *__restrict__ bl = perm + a.max_sec;                            // boundaries by position
	int *__restrict__ order = bl + 2 * a.max_sec;
	double *__restrict__ cdata = chan + 2 * a.max_sec;  // 3 * max_sec ints fit in 2 * max_sec doubles
	auto CH = [&](int k, int j) -> double & { return cdata[coff[k] + (j - clo[k])]; };
	auto chv = [&](int k, int j) -> double {  // value of the full-length row
		const int d = j - clo[k];
		return (d >= 0 && d < clen[k]) ? cdata[coff[k] + d] : 0.0;
	};
	int ns = 0;
	if (PHASE == 0) {
	// searchF0Base (reference :254-272) was evaluated by hv_unreliable_kernel
	// fixStep1 (reference :277-291; entries the reference never writes are 0)
	for (int i0 = lane; i0 < L; i0 += 256) {  // (four batches of 64 frames per trip, their loads requested together)
		double b0[4], b1[4], b2[4];
#pragma unroll
		for (int b = 0; b < 4; ++b) {
			const int i = i0 + 64 * b;
			b0[b] = i < L ? base[i] : 0.0;
			b1[b] = (i >= 2 && i < L) ? base[i - 1] : 0.0;
			b2[b] = (i >= 2 && i < L) ? base[i - 2] : 0.0;
		}
#pragma unroll
		for (int b = 0; b < 4; ++b) {
			const int i = i0 + 64 * b;
			double v = 0.0;
			if (i >= 2 && b0[b] != 0.0) {
				double ref = b1[b] * 2 - b2[b];
				v = (fabs((b0[b] - ref) / ref) > 0.008 && fabs((b0[b] - b1[b])) / b1[b] > 0.008) ? 0.0 : b0[b];
			}
			if (i < L) {
				s1[i] = v;
				s2[i] = v;
			}
		}
	}
	wave_sync();
	// fixStep2 (reference :319-334)
	ns = hv_sections(s1, L, sec, a.max_sec, lane);
	wave_sync();
	ns = min(ns, a.max_sec);
	for (int k = lane; k < ns; k += 64) {  // (a lane per section: the short ones have at most six frames to clear)
		const int st = sec[2 * k], ed = sec[2 * k + 1];
		if (ed - st >= 6) continue;
		for (int j = st; j <= ed; ++j) s2[j] = 0.0;
	}
	wave_sync();
	// fixStep3 (reference :560-585)
	for (int i0 = lane; i0 < L; i0 += 256) {
		double v[4];
#pragma unroll
		for (int b = 0; b < 4; ++b) v[b] = i0 + 64 * b < L ? s2[i0 + 64 * b] : 0.0;
#pragma unroll
		for (int b = 0; b < 4; ++b) if (i0 + 64 * b < L) s3[i0 + 64 * b] = v[b];
	}
	ns = hv_sections(s2, L, sec, a.max_sec, lane);
	wave_sync();
	ns = min(ns, a.max_sec);
	{
		int off = 0;
		for (int k = 0; k < ns; ++k) {
			const int st = sec[2 * k], ed = sec[2 * k + 1];
			const int lo = max(0, st - 104), hi = min(L - 1, ed + 104);
			if (lane == 0) { clo[k] = lo; clen[k] = hi - lo + 1; coff[k] = off; perm[k] = k; bl[2 * k] = st; bl[2 * k + 1] = ed; }
			// getMultiChannelF0 (reference :542-555)
			for (int j = lo + lane; j <= hi; j += 64) cdata[off + (j - lo)] = (j >= st && j <= ed) ? s2[j] : 0.0;
			off += hi - lo + 1;
		}
	}
	if (lane == 0) a.nsec[ui] = ns;
	return;
	}
	ns = a.nsec[ui];
	if (PHASE == 1) {
	// extend (reference :427-458) with extendF0 (:371-403)
	for (int pair = blockIdx.x; pair < 2 * ns; pair += gridDim.x) {
		const int k = pair >> 1;
		{
			const int dir = pair & 1;
			const int shift = dir == 0 ? 1 : -1;
			const int origin = dir == 0 ? bl[2 * k + 1] : bl[2 * k];
			const int last_point = dir == 0 ? min(L - 2, origin + 100) : max(1, origin - 100);
			double tmp_f0 = CH(k, origin);
			int shifted_origin = origin;
			const int distance = abs(last_point - origin);
			int miss = 0;
			// the rows the walk will visit are known in advance: request the next one while this one is searched
			auto row = [&](int t, double &c0, double &c1) {
				const double *__restrict__ r = cand + (long long)(origin + shift * t + shift) * nc;
				c0 = (t <= distance && lane < nc) ? r[lane] : 0.0;
				c1 = (t <= distance && lane + 64 < nc) ? r[lane + 64] : 0.0;
			};
			double n0, n1;
			row(0, n0, n1);
			for (int t = 0; t <= distance; ++t) {
				const int idx = origin + shift * t + shift;
				const double c0 = n0, c1 = n1;
				row(t + 1, n0, n1);
				const double sel = nc <= 128 ? hv_select_best_regs(tmp_f0, c0, c1, nc, 0.18, lane)
										 : hv_select_best(tmp_f0, cand + (long long)idx * nc, nc, 0.18, lane);
				if (lane == 0) CH(k, idx) = sel;
				if (sel == 0.0) {
					miss++;
				} else {
					tmp_f0 = sel;
					miss = 0;
					shifted_origin = idx;
				}
				if (miss == 4) break;
			}
			if (lane == 0) bl[dir == 0 ? 2 * k + 1 : 2 * k] = shifted_origin;
		}
	}
	return;
	}
	// extendSub: keep the sections that are long enough for their mean F0 (reference :441-455; mean_f0 is
	// deliberately not reset between sections)
	int count = 0;
	{
		double mean_f0 = 0.0;
		int m_st = 0, m_ed = 0, m_at = 0;  // lane l: boundaries and row address of section k0 + l (the swaps below touch positions <= k only)
		for (int k = 0; k < ns; ++k) {
			if ((k & 63) == 0) {
				const int kk = k + lane;
				if (kk < ns) {
					const int ch = perm[kk];
					m_st = bl[2 * kk]; m_ed = bl[2 * kk + 1];
					m_at = coff[ch] + (m_st - clo[ch]);
				}
			}
			const int st = __shfl(m_st, k & 63, 64), ed = __shfl(m_ed, k & 63, 64);
			mean_f0 = hv_ordered_sum(cdata + __shfl(m_at, k & 63, 64), ed - st, mean_f0, lane);
			mean_f0 /= ed - st;
			if (2200.0 / mean_f0 < ed - st) {
				wave_sync();
				if (lane == 0) {  // swapArray (reference :409-422)
					int t = perm[count]; perm[count] = perm[k]; perm[k] = t;
					t = bl[2 * count]; bl[2 * count] = bl[2 * k]; bl[2 * k] = t;
					t = bl[2 * count + 1]; bl[2 * count + 1] = bl[2 * k + 1]; bl[2 * k + 1] = t;
				}
				wave_sync();
				count++;
			}
		}
	}
	// mergeF0 (reference :502-536) with mergeF0Sub (:475-497)
	if (ns > 0) {
		// merged = the row at position 0, whatever its rank in time
		const int ch0 = perm[0];
		{
			const int lo0 = clo[ch0], len0 = clen[ch0];
			const double *__restrict__ row0 = cdata + coff[ch0];
			for (int i0 = lane; i0 < L; i0 += 256) {  // (four batches of 64 frames per trip; the row is zero outside its window)
				double v[4];
#pragma unroll
				for (int b = 0; b < 4; ++b) {
					const int d = i0 + 64 * b - lo0;
					v[b] = (d >= 0 && d < len0) ? row0[d] : 0.0;
				}
#pragma unroll
				for (int b = 0; b < 4; ++b) if (i0 + 64 * b < L) s3[i0 + 64 * b] = v[b];
			}
		}
		// order[] = positions 0..count-1 sorted by start frame the way the reference's std::sort leaves them: sections
		// that start on the same frame (several can extend back to frame 0) keep libstdc++'s order, see wc_argsort.hpp
		if (lane == 0) {
			for (int k = 0; k < count; ++k) order[k] = k;
			wc_argsort::sort_like_libstdcxx(order, count, wc_argsort::ByKey{bl, 2});
		}
		wave_sync();
		for (int q = 1; q < count; ++q) {
			const int p = order[q];
			const int ch = perm[p];
			const int i1 = bl[2 * p], i2 = bl[2 * p + 1];
			if (i1 - bl[1] > 0) {
				for (int j = i1 + lane; j <= i2; j += 64) s3[j] = chv(ch, j);
				wave_sync();
				if (lane == 0) { bl[0] = i1; bl[1] = i2; }
				wave_sync();
			} else {
				const int st1 = bl[0], ed1 = bl[1], st2 = i1, ed2 = i2;
				int new_ed = ed1;
				if (!(st1 <= st2 && ed1 >= ed2)) {
					double sc1 = 0.0, sc2 = 0.0;
					for (int j = st2 + lane; j <= ed1; j += 64) {
						double q1, q2;
						hv_search_score2(s3[j], chv(ch, j), cand + (long long)j * nc, score + (long long)j * nc, nc, q1, q2);
						sc1 += q1;
						sc2 += q2;
					}
#pragma unroll
					for (int o = 32; o > 0; o >>= 1) { sc1 += __shfl_xor(sc1, o, 64); sc2 += __shfl_xor(sc2, o, 64); }
					wave_sync();
					if (sc1 > sc2) { for (int j = ed1 + lane; j <= ed2; j += 64) s3[j] = chv(ch, j); }
					else { for (int j = st2 + lane; j <= ed2; j += 64) s3[j] = chv(ch, j); }
					new_ed = ed2;
				}
				wave_sync();
				if (lane == 0) bl[1] = new_ed;
				wave_sync();
			}
		}
	}
	wave_sync();
	// fixStep4 (reference :590-614)
	for (int i0 = lane; i0 < L; i0 += 256) {
		double v[4];
#pragma unroll
		for (int b = 0; b < 4; ++b) v[b] = i0 + 64 * b < L ? s3[i0 + 64 * b] : 0.0;
#pragma unroll
		for (int b = 0; b < 4; ++b) if (i0 + 64 * b < L) s4[i0 + 64 * b] = v[b];
	}
	ns = hv_sections(s3, L, sec, a.max_sec, lane);
	wave_sync();
	ns = min(ns, a.max_sec);
	for (int k = lane; k + 1 < ns; k += 64) {  // (a lane per gap: the ones that are filled have at most eight frames)
		const int e0 = sec[2 * k + 1], b1 = sec[2 * (k + 1)];
		const int distance = b1 - e0 - 1;
		if (distance >= 9) continue;
		const double t0 = s3[e0] + 1, t1 = s3[b1] - 1;
		const double coef = (t1 - t0) / (distance + 1.0);
		for (int j = e0 + 1; j <= b1 - 1; ++j) s4[j] = t0 + coef * (j - e0);
	}
}

// ------------------------------------------------------------------------------------------------
// zero-lag Butterworth smoothing per voiced section (reference :639-703), one lane per section
// ------------------------------------------------------------------------------------------------
struct SmArgs {
	const HvUtt *utts;
	const double *fixed;
	double *f0_1ms;
	int *sec;         // [utt][2 * max_sec]
	double *scratch;  // [utt][(L1max + 600) * 64]
	long long scratch_stride;
	int max_sec;
	int full_walk;    // WC_HARVEST_SMOOTH=full: every step of both passes over the whole padded contour (A/B and the bit-identity test)
};

constexpr int SM_PF = 32;  // steps whose inputs are requested ahead of the dependent recursion (8: 2.9 ms per half batch, memory latency per block)
__global__ __launch_bounds__(64) void hv_smooth_kernel(SmArgs a) {
	const HvUtt u = a.utts[blockIdx.x];
	const int lane = threadIdx.x;
	const int L = u.L1, lag = 300, n = L + 2 * lag;
	const double *__restrict__ f0 = a.fixed + u.l1_off;
	double *__restrict__ out = a.f0_1ms + u.l1_off;
	int *__restrict__ sec = a.sec + (long long)blockIdx.x * 2 * a.max_sec;
	double *__restrict__ tmp = a.scratch + (long long)blockIdx.x * a.scratch_stride;
	for (int i = lane; i < L; i += 64) out[i] = 0.0;
	// sections of the padded contour: padding is unvoiced, so they are the sections of f0 with the ends
	// (frames 0 and L-1) allowed to be voiced -- getBoundaryList on the padded array forces only ITS ends to 0
	int nb = 0;
	for (int base = 1; base < n; base += 256) {  // (four batches of 64 frames per trip, their loads requested together)
		double c[4], p[4];
#pragma unroll
		for (int b = 0; b < 4; ++b) {
			const int fq = base + 64 * b + lane - lag;
			c[b] = (fq >= 0 && fq < L) ? f0[fq] : 0.0;
			p[b] = (fq - 1 >= 0 && fq - 1 < L) ? f0[fq - 1] : 0.0;
		}
#pragma unroll
		for (int b = 0; b < 4; ++b) {
			const int i = base + 64 * b + lane;
			const int cur = (i >= 1 && i < n - 1 && c[b] > 0) ? 1 : 0, prv = (i - 1 >= 1 && i - 1 < n - 1 && p[b] > 0) ? 1 : 0;
			const bool ch = (i < n) && (cur != prv);
			const unsigned long long m = __ballot(ch);
			if (ch) {
				int k = nb + __popcll(m & ((1ull << lane) - 1ull));
				if (k < 2 * a.max_sec) sec[k] = i - (k & 1);
			}
			nb += __popcll(m);
		}
	}
	__syncthreads();
	const int ns = min(nb / 2, a.max_sec);
	const double b0 = 0.0078202080334971724, b1 = 0.015640416066994345;
	const double a0 = 1.7347257688092754, a1 = -0.76600660094326412;
	if (a.full_walk) {
		for (int s0 = 0; s0 < ns; s0 += 64) {
			const int k = s0 + lane;
			if (k < ns) {
				const int st = sec[2 * k], ed = sec[2 * k + 1];
				const double xs = f0[st - lag], xe = f0[ed - lag];
				// forward pass; outputs before the section start are never read back.  Loads are issued SM_PF steps
				// ahead of the dependent recursion.
				double w0 = 0.0, w1 = 0.0;
				for (int i0 = 0; i0 < n; i0 += SM_PF) {
					double xin[SM_PF];
	#pragma unroll
					for (int e = 0; e < SM_PF; ++e) xin[e] = f0[clampi(i0 + e - lag, 0, L - 1)];
	#pragma unroll
					for (int e = 0; e < SM_PF; ++e) {
						const int i = i0 + e;
						if (i < n) {
							const double xv = (i < st) ? xs : (i > ed ? xe : xin[e]);
							const double wt = xv + a0 * w0 + a1 * w1;
							if (i >= st) tmp[(long long)(n - i - 1) * 64 + lane] = b0 * wt + b1 * w0 + b0 * w1;
							w1 = w0; w0 = wt;
						}
					}
				}
				// backward pass over the reversed signal, up to the section start
				w0 = w1 = 0.0;
				const int kend = n - 1 - st;
				for (int i0 = 0; i0 <= kend; i0 += SM_PF) {
					double tin[SM_PF];
	#pragma unroll
					for (int e = 0; e < SM_PF; ++e) tin[e] = (i0 + e <= kend) ? tmp[(long long)(i0 + e) * 64 + lane] : 0.0;
	#pragma unroll
					for (int e = 0; e < SM_PF; ++e) {
						const int i = i0 + e;
						if (i <= kend) {
							const double wt = tin[e] + a0 * w0 + a1 * w1;
							const double yv = b0 * wt + b1 * w0 + b0 * w1;
							w1 = w0; w0 = wt;
							const int j = n - i - 1;
							if (j <= ed) out[j - lag] = yv;
						}
					}
				}
			}
			__syncthreads();
		}
		return;
	}
	// The reference runs both passes over the whole padded contour for every section (2 x 10 600 dependent steps for a 10 s
	// utterance).  Outside the section the input is constant -- the section's first value before it, its last value behind it
	// (:646-647) -- and the recursion is a deterministic map of its two state words, so once the state after a step equals,
	// bit for bit, the state eight steps earlier, every further block of eight steps under that input leaves it unchanged and
	// can be skipped.  (The filter settles within ~300 steps: exactly in three cases of four, else into a last-bit limit cycle
	// of period 4 or 8 -- which is why whole periods are skipped, and why the comparison is on the bits, not on a tolerance.
	// A state that never repeats -- NaN input -- simply walks every step, as the reference does.)
	//   forward:  constant lead-in (skipped up to a multiple of 8), the section, then the constant tail until its outputs
	//             repeat with period 8 (index `pend`: outputs beyond it are copies of the last eight);
	//   backward: the periodic tail from the far end (periodic input, same argument with the phase kept), the explicit tail,
	//             the section.
	// Every step that is executed is the reference's own arithmetic, in its order.
	__shared__ double per[8 * 64];  // the last period of the forward tail, [phase][lane]
	auto bits = [](double v) { return __double_as_longlong(v); };
	for (int s0 = 0; s0 < ns; s0 += 64) {
		const int k = s0 + lane;
		if (k < ns) {
			const int st = sec[2 * k], ed = sec[2 * k + 1];
			const double xs = f0[st - lag], xe = f0[ed - lag];
			double w0 = 0.0, w1 = 0.0;
			auto step = [&](double xv) -> double {
				const double wt = xv + a0 * w0 + a1 * w1;
				const double yv = b0 * wt + b1 * w0 + b0 * w1;
				w1 = w0; w0 = wt;
				return yv;
			};
			// ---- forward, lead-in: i = 0 .. st - 1 under the constant xs ----
			// (whole periods of eight steps between the comparisons, nothing else in the loop: a lone wavefront issues one
			// instruction per ~5 cycles, and a step's own dependent chain -- multiply, add, add -- is ~40; the loop counter,
			// the test of i & 7 and the divergent loop exit per step tripled that.  Where the comparison falls does not matter
			// to the result: skipping whole periods of a state that repeats leaves it as it is.)
			auto advance = [&](double xv) {  // the state alone (no output wanted)
				const double wt = xv + a0 * w0 + a1 * w1;
				w1 = w0; w0 = wt;
			};
			{
				int i = 0;
				double p0 = w0, p1 = w1;
				bool periodic = false;
				while (i + 8 <= st) {
#pragma unroll
					for (int e = 0; e < 8; ++e) advance(xs);
					i += 8;
					if (bits(w0) == bits(p0) && bits(w1) == bits(p1)) { periodic = true; break; }
					p0 = w0; p1 = w1;
				}
				if (periodic) i += ((st - i) / 8) * 8;
				for (; i < st; ++i) advance(xs);
			}
			// ---- forward, section: outputs to tmp[i - st] ----
			for (int i0 = st; i0 <= ed; i0 += SM_PF) {
				double xin[SM_PF];
#pragma unroll
				for (int e = 0; e < SM_PF; ++e) xin[e] = f0[clampi(i0 + e - lag, 0, L - 1)];
				if (__all(i0 + SM_PF - 1 <= ed)) {  // (whole blocks without the per-step test: all but each lane's last)
					double *__restrict__ tp = tmp + (long long)(i0 - st) * 64 + lane;
#pragma unroll
					for (int e = 0; e < SM_PF; ++e) tp[e * 64] = step(xin[e]);
				} else {
#pragma unroll
					for (int e = 0; e < SM_PF; ++e) {
						const int i = i0 + e;
						if (i <= ed) tmp[(long long)(i - st) * 64 + lane] = step(xin[e]);
					}
				}
			}
			// ---- forward, tail: constant xe until the outputs repeat with period 8 (or the array ends) ----
			int pend = ed + 1;
			{
				double q0 = w0, q1 = w1;
				bool settled = false;
				double *__restrict__ tp = tmp + (long long)(pend - st) * 64 + lane;
				while (pend + 8 <= n) {
#pragma unroll
					for (int e = 0; e < 8; ++e) tp[e * 64] = step(xe);
					tp += 8 * 64;
					pend += 8;
					if (bits(w0) == bits(q0) && bits(w1) == bits(q1)) { settled = true; break; }
					q0 = w0; q1 = w1;
				}
				if (!settled)
					for (; pend < n; ++pend) tmp[(long long)(pend - st) * 64 + lane] = step(xe);
			}
			// ---- backward, periodic part of the tail: j = n - 1 .. pend, input y[j] = y[pend - 8 + ((j - pend) & 7)] ----
			w0 = w1 = 0.0;
			int j = n - 1;
			if (j >= pend) {
#pragma unroll
				for (int m = 0; m < 8; ++m) per[m * 64 + lane] = tmp[(long long)(pend - 8 + m - st) * 64 + lane];
				// single steps down to a period boundary (phase 7 next), then whole periods from registers
				while (j >= pend && ((j - pend) & 7) != 7) { advance(per[((j - pend) & 7) * 64 + lane]); --j; }
				double pv[8];
#pragma unroll
				for (int m = 0; m < 8; ++m) pv[m] = per[m * 64 + lane];
				double p0 = w0, p1 = w1;
				bool periodic = false;
				while (j - 7 >= pend) {
#pragma unroll
					for (int m = 7; m >= 0; --m) advance(pv[m]);
					j -= 8;
					if (bits(w0) == bits(p0) && bits(w1) == bits(p1)) { periodic = true; break; }
					p0 = w0; p1 = w1;
				}
				if (periodic) j -= ((j - pend + 1) / 8) * 8;
				for (; j >= pend; --j) advance(per[((j - pend) & 7) * 64 + lane]);
			}
			// ---- backward, explicit tail and section: j = pend - 1 .. st ----
			for (int j0 = j; j0 >= st; j0 -= SM_PF) {
				double tin[SM_PF];
				if (__all(j0 - (SM_PF - 1) >= st)) {  // (whole blocks without the per-step tests; in the tail nothing is stored either)
					const double *__restrict__ tp = tmp + (long long)(j0 - st) * 64 + lane;
#pragma unroll
					for (int e = 0; e < SM_PF; ++e) tin[e] = tp[-(e * 64)];
					if (__all(j0 - (SM_PF - 1) > ed)) {
#pragma unroll
						for (int e = 0; e < SM_PF; ++e) advance(tin[e]);
					} else if (__all(j0 <= ed)) {
						double *__restrict__ op = out + (j0 - lag);
#pragma unroll
						for (int e = 0; e < SM_PF; ++e) op[-e] = step(tin[e]);
					} else {
#pragma unroll
						for (int e = 0; e < SM_PF; ++e) {
							const double yv = step(tin[e]);
							if (j0 - e <= ed) out[j0 - e - lag] = yv;
						}
					}
					continue;
				}
#pragma unroll
				for (int e = 0; e < SM_PF; ++e) tin[e] = (j0 - e >= st) ? tmp[(long long)(j0 - e - st) * 64 + lane] : 0.0;
#pragma unroll
				for (int e = 0; e < SM_PF; ++e) {
					const int jj = j0 - e;
					if (jj >= st) {
						const double yv = step(tin[e]);
						if (jj <= ed) out[jj - lag] = yv;
					}
				}
			}
		}
		__syncthreads();
	}
}

// reference :183-208
__global__ void hv_output_kernel(const HvUtt *__restrict__ utts, const double *__restrict__ f0_1ms, double *__restrict__ tpos,
								 double *__restrict__ f0, double frame_period) {
	const HvUtt u = utts[blockIdx.y];
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= u.L) return;
	double t, v;
	if (frame_period == 1.0) {
		t = i * 1 / 1000.0;
		v = f0_1ms[u.l1_off + i];
	} else {
		t = i * frame_period / 1000.0;
		v = f0_1ms[u.l1_off + min(u.L1 - 1, mround(t * 1000.0))];
	}
	tpos[u.out_off + i] = t;
	f0[u.out_off + i] = v;
}

}  // namespace wc

using namespace wc;

struct wc_harvest {
	int fs, decim, n_bands, S, max_cand;
	double fs_d, f0_floor, f0_ceil, frame_period, target_fs, channels_in_octave;
	Device *dev;
	std::vector<double> band_f0;
	std::vector<int> half_len, tap_off;
	DevBuf d_taps, d_tap_off, d_half_len, d_band_f0, d_ev_band_off, d_ev_cap, d_rot, d_rot8;
	DevBuf d_sd_rot, d_sd_p0, d_slot_off, d_slot_cap, slots, slot_count, seam, quiet, bmax;
	bool debug_small_caps;  // WC_DEBUG_SMALL_CAPS, read once at creation: tiny rate-bounded buffers, so that the overflow retry runs (tests)
	bool tables_valid;  // the capacity tables on the device are those of (tables_ylen, tables_full, tables_tiles)
	int tables_ylen, tables_full, tables_tiles;
	hipStream_t tables_stream = nullptr;  // the stream those uploads were enqueued on
	long long sdft8_max;  // eight lanes per (band, chunk) while that makes at most this many wavefronts (WC_HARVEST_SDFT8_MAX)
	int sdft_lanes;  // WC_HARVEST_SDFT_LANES=1 / 8: lanes per (band, chunk) of the sliding band-pass (default 0: eight for small batches); 7 / 9: the same with the detectors at every step (rounds 3-5; A/B and the bit-identity test)
	bool use_fir;  // WC_HARVEST_BANDPASS=fir: the direct FIR band-pass instead of the sliding DFT (A/B and tests)
	bool use_cos_table;  // HarvestOption::use_cos_table
	DevBuf d_cos_table;
	int phases = 3;  // hv_set_phases: 1 = front (decimation .. refinement), 2 = tail (unreliable-candidate test .. output), 3 = both
	bool no_quiet = false;  // WC_HARVEST_QUIET=sliding: no chunk is left to the FIR sums (A/B and tests)
	bool ignore_ties = false;  // WC_HARVEST_TIES=ignore: the tie flag is not acted upon (A/B and tests)
	int force_tie = -1;        // WC_HARVEST_FORCE_TIE=u (test hook, read at creation): utterance u of every stage call counts as flagged
	wc_harvest *exact_twin = nullptr;  // the same options with the band-pass as a direct FIR sum: re-runs of batches that raised the tie flag (hv_exact_twin)
	int use_cos_table_opt = 0;
	int raw_mode = 0;       // WC_HARVEST_RAW=blocks: every block of frames works out its own slice (1, round 5); =four: slices from hv_rawdesc_kernel, four wavefronts per block (2); default 0: one wavefront per block (hv_raw_wave_kernel)
	bool raw_from_lists;    // WC_HARVEST_RAW=lists: the edges packed into per-band lists before hv_raw reads them (A/B and the bit-identity test)
	long long slots_per_utt = 0;
	int refine_mode;        // WC_HARVEST_REFINE=slots: one wavefront per candidate slot (1); =packed: one wavefront per frame (2); default: frames in groups (0) (A/B and the bit-identity tests)
	bool smooth_full_walk;  // WC_HARVEST_SMOOTH=full: the smoothing filter without the skipping of settled stretches (A/B and the bit-identity test)
	bool direct_decimation;  // WC_HARVEST_DECIMATE=direct: every lane reads its own stream from memory (A/B and the bit-identity test)
	bool chunked_decimation = false;  // WC_HARVEST_DECIMATE=chunks (or direct): the chunked kernels with warm-ups of rounds 1-5 instead of exact chunk states
	DecScan dec_scan;
	DevBuf d_dec_ptab;
	DevBuf utts, dec, y, events, ev_count, overflow, tile_run, raw, cand0, cand1, score1, cand2, score2;
	DevBuf base, s1, s2, s3, fixed, f0_1ms, sec, chan, smooth, ibuf, rawdesc;
	DevBuf d_x, d_tpos, d_f0;
	HostBuf h_stage, h_x;  // (h_x: the samples of a host-pointer call on their way up)
	std::vector<HvUtt> last_utts;
	RefArgs last_refine;       // arguments of the most recent refinement launch (wc_harvest_debug_refine)
	bool last_refine_valid = false;
};

static DecCoef dec_coef(int r) {
	// FilterForDecimate coefficient sets, reference src/world_matlabfunctions.cpp:27-103
	static const double A[13][3] = {
		{0, 0, 0}, {0, 0, 0},
		{0.041156734567757189, -0.42599112459189636, 0.041037215479961225},
		{0.95039378983237421, -0.67429146741526791, 0.15412211621346475},
		{1.4499664446880227, -0.98943497080950582, 0.24578252340690215},
		{1.7610939654280557, -1.2554914843859768, 0.3237186507788215},
		{1.9715352749512141, -1.4686795689225347, 0.3893908434965701},
		{2.1225239019534703, -1.6395144861046302, 0.44469707800587366},
		{2.2357462340187593, -1.7780899984041358, 0.49152555365968692},
		{2.3236003491759578, -1.8921545617463598, 0.53148928133729068},
		{2.3936475118069387, -1.9873904075111861, 0.5658879979027055},
		{2.450743295230728, -2.06794904601978, 0.59574774438332101},
		{2.4981398605924205, -2.1368928194784025, 0.62187513816221485}};
	static const double B[13][2] = {
		{0, 0}, {0, 0},
		{0.16797464681802227, 0.50392394045406674},
		{0.071221945171178636, 0.21366583551353591},
		{0.036710750339322612, 0.11013225101796784},
		{0.021334858522387423, 0.06400457556716227},
		{0.013469181309343825, 0.040407543928031475},
		{0.0090366882681608418, 0.027110064804482525},
		{0.0063522763407111993, 0.019056829022133598},
		{0.0046331164041389372, 0.013899349212416812},
		{0.0034818622251927556, 0.010445586675578267},
		{0.0026822508007163792, 0.0080467524021491377},
		{0.0021097275904709001, 0.0063291827714127002}};
	int i = (r >= 2 && r <= 12) ? r : 0;
	return DecCoef{A[i][0], A[i][1], A[i][2], B[i][0], B[i][1]};
}

static inline int h_mround(double x) { return x > 0 ? static_cast<int>(x + 0.5) : static_cast<int>(x - 0.5); }

// Enqueue-only (no host synchronisation), shared with the fused pipeline.  `full` selects the hard bound for the
// zero-crossing buffers; with the rate bound an overflow is reported through h->overflow (device) and handled by
// the caller (hv_overflowed) by re-running with full == true.
// mode: 0 = frames in groups (hv_refine_group_kernel; rows wider than 128 positions take the packed kernel), 1 = slots, 2 = packed
static void launch_refine(const RefArgs &fa, hipStream_t s, int mode, bool table) {
	const unsigned frames = (unsigned)(8 * ((fa.total_frames + 7) / 8));  // (the packed kernel deals frames to the XCDs in eighths)
	const bool small = 7 * fa.p.S <= 112;
	if (mode == 1) {
		if (table) hipLaunchKernelGGL(hv_refine_kernel<true>, dim3(frames), dim3(256), 0, s, fa);
		else hipLaunchKernelGGL(hv_refine_kernel<false>, dim3(frames), dim3(256), 0, s, fa);
	} else if (mode == 0 && !small && 7 * fa.p.S <= 128) {
		const int n_grp = (fa.max_l1 + WC_RQ_F - 1) / WC_RQ_F;
		const dim3 blocks((unsigned)(8 * ((n_grp + 7) / 8)), (unsigned)fa.n_utt);
		if (table) hipLaunchKernelGGL((hv_refine_group_kernel<true, WC_RQ_F, 128>), blocks, dim3(64 * WC_RQ_F), 0, s, fa);
		else hipLaunchKernelGGL((hv_refine_group_kernel<false, WC_RQ_F, 128>), blocks, dim3(64 * WC_RQ_F), 0, s, fa);
	} else if (mode == 0 && small) {
		const int n_grp = (fa.max_l1 + WC_RQ_F - 1) / WC_RQ_F;
		const dim3 blocks((unsigned)(8 * ((n_grp + 7) / 8)), (unsigned)fa.n_utt);
		if (table) hipLaunchKernelGGL((hv_refine_group_kernel<true, WC_RQ_F>), blocks, dim3(64 * WC_RQ_F), 0, s, fa);
		else hipLaunchKernelGGL((hv_refine_group_kernel<false, WC_RQ_F>), blocks, dim3(64 * WC_RQ_F), 0, s, fa);
#if WC_RQ_PROF
		{
			unsigned long long h_[16];
			hipStreamSynchronize(s);
			hipMemcpyFromSymbol(h_, HIP_SYMBOL(rq_prof), sizeof(h_));
			static const char *nm[12] = {"gather", "dedupe", "barrier1", "rank+2 barriers", "phase1", "barrier4", "grab", "pass setup", "sample loop", "closing", "members", "-"};
			unsigned long long tot = 0;
			for (int k = 0; k < 11; ++k) tot += h_[k];
			for (int k = 0; k < 11; ++k) fprintf(stderr, "rq_prof %-16s %8.1f Mcycles %5.1f %%\n", nm[k], h_[k] / 1e6, 100.0 * h_[k] / tot);
			unsigned long long z_[16] = {0};
			hipMemcpyToSymbol(HIP_SYMBOL(rq_prof), z_, sizeof(z_));
		}
#endif
	} else if (mode == 3 && small && !table) {  // (A/B: two frames per workgroup)
		const int n_grp = (fa.max_l1 + 1) / 2;
		hipLaunchKernelGGL((hv_refine_group_kernel<false, 2>), dim3((unsigned)(8 * ((n_grp + 7) / 8)), (unsigned)fa.n_utt), dim3(128), 0, s, fa);
	} else if (mode == 4 && small && !table) {  // (A/B: eight frames per workgroup)
		const int n_grp = (fa.max_l1 + 7) / 8;
		hipLaunchKernelGGL((hv_refine_group_kernel<false, 8>), dim3((unsigned)(8 * ((n_grp + 7) / 8)), (unsigned)fa.n_utt), dim3(512), 0, s, fa);
	} else {
		if (table) {
			if (small) hipLaunchKernelGGL((hv_refine_packed_kernel<true, 112>), dim3(frames), dim3(64), 0, s, fa);
			else hipLaunchKernelGGL((hv_refine_packed_kernel<true, 7 * MAX_SLOTS>), dim3(frames), dim3(64), 0, s, fa);
		} else {
			if (small) hipLaunchKernelGGL((hv_refine_packed_kernel<false, 112>), dim3(frames), dim3(64), 0, s, fa);
			else hipLaunchKernelGGL((hv_refine_packed_kernel<false, 7 * MAX_SLOTS>), dim3(frames), dim3(64), 0, s, fa);
		}
	}
}

// part: 3 = the whole chain; 1 = the front only (decimation .. refinement); 2 = the tail of a chain whose front an earlier call
// with the same arguments has enqueued on the same stream (nothing is uploaded or cleared again).  bp_done: recorded behind the
// band-pass; tail_after: the tail waits for it.  (The pipeline's two staggered chains: the first chain's tail -- a few long-lived
// wavefronts -- starts when the second chain's band-pass is through, see wc_pipeline.hip.)
int hv_enqueue(wc_harvest *h, hipStream_t s, int n_utt, const double *d_x, const int *x_length, double *d_tpos, double *d_f0,
			   bool full, hipEvent_t mid_event, hipEvent_t start_after, int part, hipEvent_t bp_done, hipEvent_t tail_after) {
	Device *dev = h->dev;
	// part, bits 0 / 1: front / tail.  Bit 2 (4): of the front only what needs nothing but the samples -- decimation, DC, the level
	// marks and the seam values in front of the band-pass --; bit 3 (8): the front behind that (an earlier call with bit 2 has
	// enqueued the rest, maybe on another stream: `start_after` orders the two).  Round 6: the pipeline's two-lane schedule puts the
	// second group's decimation on its latency lane, underneath the first group's band-pass.
	const bool pre_only = (part & 4) != 0, main_only = (part & 8) != 0;
	const int phases = h->phases & part & 3;
	const bool resume_tail = (part & 3) == 2;
	const bool resume = resume_tail || main_only;  // (nothing is uploaded or cleared again)
	const int r = h->decim;
	const int lag = (r == 1) ? 0 : static_cast<int>(std::ceil(140.0 / r) * r);
	std::vector<HvUtt> utts(n_utt);
	long long xo = 0, deco = 0, yo = 0, l1o = 0, oo = 0;
	int max_len = 0, max_ylen = 0, max_L1 = 0, max_L = 0;
	for (int u = 0; u < n_utt; ++u) {
		if (x_length[u] <= 0) return fail(WC_ERR_INVALID, "harvest: non-positive x_length");
		HvUtt &t = utts[u];
		t.x_off = xo; t.dec_off = deco; t.y_off = yo + Y_PADL; t.l1_off = l1o; t.out_off = oo; t.ev_off = 0;
		t.x_len = x_length[u];
		t.y_len = 1 + x_length[u] / r;                              // reference :1400
		t.L1 = wc_get_samples(h->fs, x_length[u], 1);                // reference :1411
		t.L = (h->frame_period == 1.0) ? t.L1 : wc_get_samples(h->fs, x_length[u], h->frame_period);
		if (t.L1 < 3) return fail(WC_ERR_INVALID, "harvest: signal shorter than 3 ms");
		const int len = t.x_len + 2 * lag + 18;
		xo += t.x_len; deco += len; yo += t.y_len + Y_PADL + Y_PADR; l1o += t.L1; oo += t.L;
		max_len = std::max(max_len, len); max_ylen = std::max(max_ylen, t.y_len);
		max_L1 = std::max(max_L1, t.L1); max_L = std::max(max_L, t.L);
	}
	const long long total_l1 = l1o;
	const int nb = h->n_bands, S = h->S, nc = 7 * S;
	const int max_sec = max_L1 / 2 + 8;
	int rc;
	// event capacities per band and type: a rate bound, or the hard bound when `full`
	std::vector<long long> ev_band_off(nb);
	std::vector<int> ev_cap(nb);
	if ((rc = h->overflow.reserve((2 + (size_t)n_utt) * sizeof(int)))) return rc;  // [0] overflow, [1] tie, [2 + u] tie in utterance u (RefArgs::flags)
	if ((rc = h->utts.reserve(sizeof(HvUtt) * n_utt))) return rc;
	if ((rc = h->y.reserve(sizeof(double) * yo))) return rc;
	if (r != 1 && (rc = h->dec.reserve(sizeof(double) * deco))) return rc;
	if ((rc = h->ev_count.reserve(sizeof(int) * 4ll * nb * n_utt))) return rc;
	const int tile_adv = h->use_fir ? FIR_ADV : BP_ADV;
	const int n_tiles = (max_ylen + tile_adv - 1) / tile_adv;
	if ((rc = h->tile_run.reserve(sizeof(int) * 4ll * (n_tiles + 1) * nb * n_utt))) return rc;
	if ((rc = h->raw.reserve(sizeof(double) * total_l1 * nb))) return rc;
	if ((rc = h->cand0.reserve(sizeof(double) * total_l1 * S))) return rc;
	if ((rc = h->cand1.reserve(sizeof(double) * total_l1 * nc))) return rc;
	if ((rc = h->score1.reserve(sizeof(double) * total_l1 * nc))) return rc;
	if ((rc = h->cand2.reserve(sizeof(double) * total_l1 * nc))) return rc;
	if ((rc = h->score2.reserve(sizeof(double) * total_l1 * nc))) return rc;
	for (DevBuf *b : {&h->base, &h->s1, &h->s2, &h->s3, &h->fixed, &h->f0_1ms})
		if ((rc = b->reserve(sizeof(double) * total_l1))) return rc;
	if ((rc = h->sec.reserve(sizeof(int) * 2ll * max_sec * n_utt))) return rc;
	// after fixStep2 a section spans at least 7 frames plus one unvoiced frame
	const long long chan_stride = 2ll * max_sec + (long long)max_L1 + 209ll * (max_L1 / 8 + 2) + 64;
	if ((rc = h->ibuf.reserve(sizeof(int) * (4ll * max_sec + 1) * n_utt))) return rc;
	if ((rc = h->chan.reserve(sizeof(double) * chan_stride * n_utt))) return rc;
	const long long smooth_stride = (long long)(max_L1 + 600) * 64;
	if ((rc = h->smooth.reserve(sizeof(double) * smooth_stride * n_utt))) return rc;

	{
		long long per_utt = 0;
		for (int b = 0; b < nb; ++b) {
			int hard = max_ylen / 2 + 4;
			int soft = static_cast<int>(2.5 * h->band_f0[b] * (max_ylen / h->fs_d)) + 64;
			if (h->debug_small_caps) soft = 24;  // test hook: forces the overflow-and-retry path
			ev_cap[b] = full ? hard : std::min(hard, soft);
			ev_band_off[b] = per_utt;
			per_utt += 4ll * ev_cap[b];
		}
		for (int u = 0; u < n_utt; ++u) utts[u].ev_off = per_utt * u;
		// (per-band edge lists: only the FIR band-pass and WC_HARVEST_RAW=lists use them -- by default hv_raw reads the band-pass's slots)
		if ((h->use_fir || h->raw_from_lists) && (rc = h->events.reserve(sizeof(double) * per_utt * n_utt))) return rc;
		if ((rc = h->d_ev_band_off.reserve(sizeof(long long) * nb))) return rc;
		if ((rc = h->d_ev_cap.reserve(sizeof(int) * nb))) return rc;
		// sliding band-pass: one slot per (band, chunk, type); same rate bound / hard bound policy per chunk
		std::vector<long long> slot_off(nb);
		std::vector<int> slot_cap(nb);
		long long slots_per_utt = 0;
		for (int b = 0; b < nb; ++b) {
			const int hard = SD_CH / 2 + 2;
			int soft = static_cast<int>(2.5 * h->band_f0[b] * (SD_CH / h->fs_d)) + 16;
			if (h->debug_small_caps) soft = 3;
			slot_cap[b] = ((full ? hard : std::min(hard, soft)) + 3) & ~3;  // (whole 32-byte sectors: the sliding band-pass writes its edges four at a time)
			slot_off[b] = slots_per_utt;
			slots_per_utt += 4ll * n_tiles * slot_cap[b];
		}
		if (!h->use_fir) {
			if ((rc = h->slots.reserve(sizeof(double) * slots_per_utt * n_utt))) return rc;
			if ((rc = h->slot_count.reserve(sizeof(int) * 4ll * n_tiles * nb * n_utt))) return rc;
			if ((rc = h->seam.reserve(sizeof(double) * 2ll * (n_tiles + 1) * nb * n_utt))) return rc;
			if ((rc = h->quiet.reserve(sizeof(int) * ((long long)n_tiles + 1) * n_utt))) return rc;
			if ((rc = h->bmax.reserve(sizeof(double) * (long long)((max_ylen + 63) / 64) * n_utt))) return rc;
			if ((rc = h->d_slot_off.reserve(sizeof(long long) * nb))) return rc;
			if ((rc = h->d_slot_cap.reserve(sizeof(int) * nb))) return rc;
		}
		const size_t o1 = sizeof(HvUtt) * n_utt, o2 = o1 + sizeof(long long) * nb, o3 = o2 + sizeof(int) * nb + 8;
		const size_t o3a = o3 & ~size_t(7), o4 = o3a + sizeof(long long) * nb;
		char *hs = nullptr;
		if (!resume) {
			if ((rc = h->h_stage.reserve(o4 + sizeof(int) * nb + 64))) return rc;
			hs = static_cast<char *>(h->h_stage.p);
			std::memcpy(hs, utts.data(), sizeof(HvUtt) * n_utt);
			std::memcpy(hs + o1, ev_band_off.data(), sizeof(long long) * nb);
			std::memcpy(hs + o2, ev_cap.data(), sizeof(int) * nb);
			std::memcpy(hs + o3a, slot_off.data(), sizeof(long long) * nb);
			std::memcpy(hs + o4, slot_cap.data(), sizeof(int) * nb);
		}
		if (!resume) WC_HIP(hipMemcpyAsync(h->utts.p, hs, sizeof(HvUtt) * n_utt, hipMemcpyHostToDevice, s));
		// the per-band capacity tables depend on the longest utterance and the retry flag only: a call like the one before (the usual
		// case of a stream of equal-sized batches) finds them on the device already -- four copies less in front of the first kernel
		// (on the SAME stream: the uploads are only ordered against kernels behind them on the stream they were enqueued on, and
		// wc_set_stream / the pipeline may hand this handle another one from call to call)
		const bool same_tables = h->tables_valid && h->tables_stream == s && h->tables_ylen == max_ylen && h->tables_full == (full ? 1 : 0) &&
								 h->tables_tiles == n_tiles;
		if (!same_tables && !resume) {
			h->tables_valid = false;  // (valid again only once all four copies have been enqueued)
			WC_HIP(hipMemcpyAsync(h->d_ev_band_off.p, hs + o1, sizeof(long long) * nb, hipMemcpyHostToDevice, s));
			WC_HIP(hipMemcpyAsync(h->d_ev_cap.p, hs + o2, sizeof(int) * nb, hipMemcpyHostToDevice, s));
			if (!h->use_fir) {
				WC_HIP(hipMemcpyAsync(h->d_slot_off.p, hs + o3a, sizeof(long long) * nb, hipMemcpyHostToDevice, s));
				WC_HIP(hipMemcpyAsync(h->d_slot_cap.p, hs + o4, sizeof(int) * nb, hipMemcpyHostToDevice, s));
			}
			h->tables_valid = true; h->tables_stream = s; h->tables_ylen = max_ylen; h->tables_full = full ? 1 : 0; h->tables_tiles = n_tiles;
		}
		if (!resume) {
			if ((rc = h->h_stage.mark(s))) return rc;
			WC_HIP(hipMemsetAsync(h->overflow.p, 0, (2 + (size_t)n_utt) * sizeof(int), s));
		}
		const HvUtt *du = h->utts.as<HvUtt>();
		if ((phases & 1) && !main_only) {
			WC_HIP(hipMemsetAsync(h->y.p, 0, sizeof(double) * yo, s));
			if ((rc = dev->time_begin("harvest_decimate", s))) return rc;
			if (r == 1) {
				hipLaunchKernelGGL(hv_copy_kernel, dim3((max_ylen + 255) / 256, n_utt), dim3(256), 0, s, du, d_x, h->y.as<double>());
			} else {
				const DecCoef c = dec_coef(r);
				const int chunks = (max_len + DEC_CHUNK - 1) / DEC_CHUNK;
				dim3 grid((chunks + 63) / 64, n_utt);
				if (!h->chunked_decimation) {
					const dim3 sgrid((max_len + DS_SPAN - 1) / DS_SPAN, n_utt);
					hipLaunchKernelGGL(hv_decimate_scan_kernel<0>, sgrid, dim3(DS_T), 0, s, du, d_x, h->dec.as<double>(), h->y.as<double>(), c, h->dec_scan, h->d_dec_ptab.as<double>(), r, lag);
					hipLaunchKernelGGL(hv_decimate_scan_kernel<1>, sgrid, dim3(DS_T), 0, s, du, d_x, h->dec.as<double>(), h->y.as<double>(), c, h->dec_scan, h->d_dec_ptab.as<double>(), r, lag);
				} else if (h->direct_decimation) {
					hipLaunchKernelGGL(hv_decimate_kernel<0>, grid, dim3(64), 0, s, du, d_x, h->dec.as<double>(), h->y.as<double>(), c, r, lag);
					hipLaunchKernelGGL(hv_decimate_kernel<1>, grid, dim3(64), 0, s, du, d_x, h->dec.as<double>(), h->y.as<double>(), c, r, lag);
				} else {
					hipLaunchKernelGGL(hv_decimate_lds_kernel<0>, grid, dim3(64), 0, s, du, d_x, h->dec.as<double>(), h->y.as<double>(), c, r, lag);
					hipLaunchKernelGGL(hv_decimate_lds_kernel<1>, grid, dim3(64), 0, s, du, d_x, h->dec.as<double>(), h->y.as<double>(), c, r, lag);
				}
			}
			hipLaunchKernelGGL(hv_dc_kernel, dim3(n_utt), dim3(1024), 0, s, du, h->y.as<double>());
			WC_HIP(hipGetLastError());
			if ((rc = dev->time_end("harvest_decimate", s))) return rc;
		}
		// a staggered twin chain holds its ALU-bound kernels back until the other chain's are through; the latency-bound
		// decimation above may run underneath them
		if (start_after && (phases & 1)) WC_HIP(hipStreamWaitEvent(s, start_after, 0));
		if (phases & 1) {
		BpArgs ba;
		ba.utts = du; ba.y = h->y.as<double>(); ba.taps = h->d_taps.as<double>(); ba.tap_off = h->d_tap_off.as<int>();
		ba.half_len = h->d_half_len.as<int>(); ba.ev_band_off = h->d_ev_band_off.as<long long>(); ba.ev_cap = h->d_ev_cap.as<int>();
		ba.events = h->events.as<double>(); ba.ev_count = h->ev_count.as<int>(); ba.overflow = h->overflow.as<int>(); ba.n_bands = nb;
		ba.tile_run = h->tile_run.as<int>(); ba.n_tiles = n_tiles;
		if (!pre_only && (rc = dev->time_begin("harvest_bandpass", s))) return rc;
		if (h->use_fir) {
			if (pre_only) { h->last_utts = utts; return WC_OK; }
			hipLaunchKernelGGL(hv_bandpass_kernel, dim3(nb, n_utt), dim3(BP_T), 0, s, ba);
		} else {
			SdArgs sa;
			sa.utts = du; sa.y = h->y.as<double>(); sa.rot = h->d_sd_rot.as<double2>(); sa.p0 = h->d_sd_p0.as<double2>();
			sa.half_len = h->d_half_len.as<int>(); sa.slot_off = h->d_slot_off.as<long long>(); sa.slot_cap = h->d_slot_cap.as<int>();
			sa.slots_per_utt = slots_per_utt; sa.slots = h->slots.as<double>(); sa.slot_count = h->slot_count.as<int>();
			h->slots_per_utt = slots_per_utt;
			sa.n_bands = nb; sa.n_chunks = n_tiles;
			sa.taps = h->d_taps.as<double>(); sa.tap_off = h->d_tap_off.as<int>(); sa.seam = h->seam.as<double>();
			sa.n_blk = (max_ylen + 63) / 64;
			sa.hl_max = *std::max_element(h->half_len.begin(), h->half_len.end());
			sa.bmax = h->bmax.as<double>(); sa.quiet = h->quiet.as<int>();
			if (!main_only) {
				hipLaunchKernelGGL(hv_blockmax_kernel, dim3((sa.n_blk + 3) / 4, n_utt), dim3(256), 0, s, sa, h->bmax.as<double>());
				hipLaunchKernelGGL(hv_quiet_kernel, dim3(n_utt), dim3(64), 0, s, sa, h->quiet.as<int>(), h->no_quiet ? 0 : 1);
				hipLaunchKernelGGL(hv_seam_kernel, dim3(nb, n_utt), dim3(64), 0, s, sa);
			}
			if (pre_only) {
				WC_HIP(hipGetLastError());
				h->last_utts = utts;
				return WC_OK;
			}
			// small batches leave most of the chip idle with a lane per (band, chunk): eight lanes each then (same bits)
			const long long waves1 = (long long)((nb * n_tiles + 63) / 64) * n_utt;
			if (h->sdft_lanes == 8 || h->sdft_lanes == 9 || (h->sdft_lanes == 0 && waves1 * 8 <= h->sdft8_max)) {
				if (h->sdft_lanes == 9)  // (the step-by-step form of the eight-lane kernel: A/B and the bit-identity test)
					hipLaunchKernelGGL(hv_bandpass_sdft8_kernel, dim3((nb * n_tiles + 7) / 8, n_utt), dim3(64), 0, s, sa);
				else
					hipLaunchKernelGGL(hv_bandpass_sdft8b_kernel, dim3((nb * n_tiles + 7) / 8, n_utt), dim3(64), 0, s, sa);
			} else if (h->sdft_lanes == 7) {  // (the one-lane kernel with its detectors at every step: A/B and the bit-identity test)
				hipLaunchKernelGGL(hv_bandpass_sdft_kernel<false>, dim3((nb * n_tiles + 63) / 64, n_utt), dim3(64), 0, s, sa);
			} else {
				hipLaunchKernelGGL(hv_bandpass_sdft_kernel<true>, dim3((nb * n_tiles + 63) / 64, n_utt), dim3(64), 0, s, sa);
			}
			if (!h->no_quiet) hipLaunchKernelGGL(hv_bandpass_quiet_kernel, dim3(nb, n_utt), dim3(BP_T), 0, s, sa);
			CpArgs ca;
			ca.utts = du; ca.slot_off = sa.slot_off; ca.slot_cap = sa.slot_cap; ca.slots_per_utt = slots_per_utt; ca.slots = sa.slots;
			ca.slot_count = sa.slot_count; ca.ev_band_off = ba.ev_band_off; ca.ev_cap = ba.ev_cap; ca.events = ba.events;
			ca.ev_count = ba.ev_count; ca.overflow = ba.overflow; ca.tile_run = ba.tile_run; ca.n_bands = nb; ca.n_chunks = n_tiles;
			ca.copy = h->raw_from_lists ? 1 : 0;
			hipLaunchKernelGGL(hv_compact_kernel, dim3(nb, n_utt), dim3(256), 0, s, ca);
		}
		WC_HIP(hipGetLastError());
		if ((rc = dev->time_end("harvest_bandpass", s))) return rc;
		if (bp_done) WC_HIP(hipEventRecord(bp_done, s));
		}
	}
	const HvUtt *du = h->utts.as<HvUtt>();
	if (phases & 1) {
	RawArgs ra;
	ra.utts = du; ra.events = h->events.as<double>(); ra.ev_band_off = h->d_ev_band_off.as<long long>(); ra.ev_cap = h->d_ev_cap.as<int>();
	ra.ev_count = h->ev_count.as<int>(); ra.band_f0 = h->d_band_f0.as<double>(); ra.raw = h->raw.as<double>(); ra.n_bands = nb;
	ra.tile_run = h->tile_run.as<int>(); ra.n_tiles = n_tiles; ra.tile_adv = tile_adv;
	ra.fs_d = h->fs_d; ra.f0_floor = h->f0_floor; ra.f0_ceil = h->f0_ceil; ra.r_fs_d = 1.0 / h->fs_d;
	if ((rc = dev->time_begin("harvest_raw", s))) return rc;
	ra.slots = h->slots.as<double>(); ra.slot_off = h->d_slot_off.as<long long>(); ra.slot_cap = h->d_slot_cap.as<int>();
	ra.slots_per_utt = h->slots_per_utt;
	if (h->use_fir || h->raw_from_lists) hipLaunchKernelGGL((hv_raw_kernel<false, false>), dim3((max_L1 + 255) / 256, nb, n_utt), dim3(256), 0, s, ra);
	else if (h->raw_mode == 1) hipLaunchKernelGGL((hv_raw_kernel<true, false>), dim3((max_L1 + 255) / 256, nb, n_utt), dim3(256), 0, s, ra);
	else {
		// the slice of every (utterance, band, block of frames, type) worked out once, a thread each, instead of by every lane of
		// the block's wavefront behind three dependent look-ups
		const int nblk = (max_L1 + RAW_T - 1) / RAW_T;
		if ((rc = h->rawdesc.reserve(sizeof(int4) * 2ll * 4 * nblk * nb * n_utt))) return rc;
		ra.desc = h->rawdesc.as<int4>();
		ra.desc_blocks = nblk;
		const long long nd = 4ll * nblk * nb * n_utt;
		hipLaunchKernelGGL(hv_rawdesc_kernel, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, s, ra, h->rawdesc.as<int4>(), nblk, n_utt);
		if (h->raw_mode == 2) hipLaunchKernelGGL((hv_raw_kernel<true, true>), dim3(nblk, nb, n_utt), dim3(256), 0, s, ra);
		else hipLaunchKernelGGL(hv_raw_wave_kernel, dim3(nblk, nb, n_utt), dim3(64), 0, s, ra);
	}
#if WC_RQ_PROF
	{
		unsigned long long h_[16];
		hipStreamSynchronize(s);
		hipMemcpyFromSymbol(h_, HIP_SYMBOL(rq_prof), sizeof(h_));
		static const char *nm[6] = {"set-up", "slice bounds", "staging", "barrier", "frames", "store"};
		unsigned long long tot = 0;
		for (int k = 0; k < 6; ++k) tot += h_[k];
		for (int k = 0; k < 6; ++k) fprintf(stderr, "raw_prof %-16s %8.1f Mcycles %5.1f %%\n", nm[k], h_[k] / 1e6, 100.0 * h_[k] / tot);
		unsigned long long z_[16] = {0};
		hipMemcpyToSymbol(HIP_SYMBOL(rq_prof), z_, sizeof(z_));
	}
#endif
	hipLaunchKernelGGL(hv_detect_kernel, dim3((max_L1 + 255) / 256, n_utt), dim3(256), 0, s, du, h->raw.as<double>(), h->cand0.as<double>(), nb, S);
	WC_HIP(hipGetLastError());
	if ((rc = dev->time_end("harvest_raw", s))) return rc;
	RefArgs fa;
	fa.utts = du; fa.n_utt = n_utt; fa.y = h->y.as<double>(); fa.cand0 = h->cand0.as<double>(); fa.tw = dev->twiddle; fa.rot = h->d_rot.as<double2>();
	fa.rot8 = h->d_rot8.as<double2>();
	fa.cand1 = h->cand1.as<double>(); fa.score1 = h->score1.as<double>(); fa.total_frames = total_l1; fa.max_l1 = max_L1;
	fa.p.fs = h->fs; fa.p.decim = r; fa.p.n_bands = nb; fa.p.S = S; fa.p.n_cand = nc; fa.p.fs_d = h->fs_d;
	fa.p.f0_floor = h->f0_floor; fa.p.f0_ceil = h->f0_ceil; fa.p.frame_period = h->frame_period;
	if ((rc = dev->time_begin("harvest_refine", s))) return rc;
	fa.cos_table = h->d_cos_table.as<double>();
	fa.flags = h->overflow.as<int>();
	launch_refine(fa, s, h->refine_mode, h->use_cos_table);
	h->last_refine = fa;
	h->last_refine_valid = true;
	WC_HIP(hipGetLastError());
	if ((rc = dev->time_end("harvest_refine", s))) return rc;
	}
	// the ALU-bound part of this chain is enqueued: a staggered twin chain may start its own now, next to our
	// latency-bound tail (contour logic, smoothing) and whatever the caller runs after us
	// Round 4: the twin chain starts behind the candidate test (hv_unreliable) as well, not right behind the refinement: that
	// full-grid kernel takes wavefront places of which the twin's band-pass -- one round of 3040 long-lived wavefronts on 3072
	// places -- wants all (1.83 ms alone, 2.55 ms beside it): 27.8 -> 27.2 ms per 64 x 10 s.  Later still (behind the contour
	// kernels, one wavefront per utterance or section) exposes their latency: 27.9 / 28.1 ms.  WC_HARVEST_MID_LATE=0..3 (A/B).
	static const int mid_late = getenv("WC_HARVEST_MID_LATE") ? atoi(getenv("WC_HARVEST_MID_LATE")) : 1;
	if (mid_event && !resume_tail && !(mid_late && (phases & 2))) WC_HIP(hipEventRecord(mid_event, s));
	if (!(phases & 2)) {
		h->last_utts = utts;
		return WC_OK;
	}
	if (tail_after) WC_HIP(hipStreamWaitEvent(s, tail_after, 0));
	if ((rc = dev->time_begin("harvest_contour", s))) return rc;  // the per-utterance tail: unreliable-candidate test, contour logic, smoothing
	const size_t unr_lds = sizeof(double) * (size_t)(3 * UNR_F + 2) * nc + sizeof(int) * (2 * UNR_F + 2) + (size_t)UNR_F * nc;
	hipLaunchKernelGGL(hv_unreliable_kernel, dim3((unsigned)((max_L1 + UNR_F - 1) / UNR_F), n_utt), dim3(256), unr_lds, s, du, h->cand1.as<double>(),
					   h->score1.as<double>(), h->cand2.as<double>(), h->score2.as<double>(), h->base.as<double>(), nc);
	if (mid_event && !resume_tail && mid_late == 1) WC_HIP(hipEventRecord(mid_event, s));
	CtrArgs ca;
	ca.utts = du; ca.cand = h->cand2.as<double>(); ca.score = h->score2.as<double>(); ca.base = h->base.as<double>();
	ca.s1 = h->s1.as<double>(); ca.s2 = h->s2.as<double>(); ca.s3 = h->s3.as<double>(); ca.fixed = h->fixed.as<double>();
	ca.sec = h->sec.as<int>(); ca.chan = h->chan.as<double>(); ca.chan_stride = chan_stride; ca.max_sec = max_sec; ca.nc = nc;
	ca.ibuf = h->ibuf.as<int>();
	ca.nsec = h->ibuf.as<int>() + 4ll * max_sec * n_utt;
	hipLaunchKernelGGL(hv_contour_kernel<0>, dim3(n_utt), dim3(64), 0, s, ca);
	if (mid_event && !resume_tail && mid_late == 2) WC_HIP(hipEventRecord(mid_event, s));
	hipLaunchKernelGGL(hv_contour_kernel<1>, dim3(96, n_utt), dim3(64), 0, s, ca);  // a 10 s utterance has 20-40 sections
	if (mid_event && !resume_tail && mid_late == 3) WC_HIP(hipEventRecord(mid_event, s));
	hipLaunchKernelGGL(hv_contour_kernel<2>, dim3(n_utt), dim3(64), 0, s, ca);
	SmArgs sa;
	sa.utts = du; sa.fixed = h->fixed.as<double>(); sa.f0_1ms = h->f0_1ms.as<double>(); sa.sec = h->sec.as<int>();
	sa.scratch = h->smooth.as<double>(); sa.scratch_stride = smooth_stride; sa.max_sec = max_sec;
	sa.full_walk = h->smooth_full_walk ? 1 : 0;
	hipLaunchKernelGGL(hv_smooth_kernel, dim3(n_utt), dim3(64), 0, s, sa);
	hipLaunchKernelGGL(hv_output_kernel, dim3((max_L + 255) / 256, n_utt), dim3(256), 0, s, du, h->f0_1ms.as<double>(), d_tpos, d_f0, h->frame_period);
	WC_HIP(hipGetLastError());
	if ((rc = dev->time_end("harvest_contour", s))) return rc;
	h->last_utts = utts;
	return WC_OK;
}

// Harvest in two parts (the incremental stream path, wc_stream.hip): a handle restricted to the front leaves the refined
// candidate and score rows of its batch ([1 ms frame][7 S], packed by utterance) in hv_candidate_rows / hv_score_rows; a handle
// restricted to the tail runs the unreliable-candidate test, the contour logic and the smoothing on rows the caller has put there
// (hv_reserve_rows first, so that hv_enqueue does not move the buffers).
void hv_set_phases(wc_harvest *h, int mask) { h->phases = mask & 3; }
int hv_row_width(const wc_harvest *h) { return 7 * h->S; }
int hv_reserve_rows(wc_harvest *h, long long total_l1) {
	int rc;
	if ((rc = h->cand1.reserve(sizeof(double) * total_l1 * 7 * h->S))) return rc;
	return h->score1.reserve(sizeof(double) * total_l1 * 7 * h->S);
}
double *hv_candidate_rows(wc_harvest *h) { return h->cand1.as<double>(); }
double *hv_score_rows(wc_harvest *h) { return h->score1.as<double>(); }

// tie statistics of the process (development / bench): utterances that went through a refinement, utterances flagged
static std::atomic<unsigned long long> g_tie_seen{0}, g_tie_flagged{0};

// after hv_enqueue: synchronises the stream and reports whether the zero-crossing buffers overflowed and (tie != NULL) whether a
// raw candidate sat on a tie of the refinement's integer decisions (never reported by a handle whose band-pass is the FIR sum);
// tie_utts: the utterances of the call in which one did (round 6: only those are run again)
int hv_overflowed(wc_harvest *h, hipStream_t s, bool *overflow, bool *tie, std::vector<int> *tie_utts) {
	const int n = (int)h->last_utts.size();
	std::vector<int> fl(2 + (size_t)n, 0);
	WC_HIP(hipMemcpyAsync(fl.data(), h->overflow.p, (tie_utts ? fl.size() : 2) * sizeof(int), hipMemcpyDeviceToHost, s));
	WC_HIP(hipStreamSynchronize(s));
	*overflow = fl[0] != 0;
	const bool acts = !h->use_fir && !h->ignore_ties;
	if (tie) *tie = fl[1] != 0 && acts;
	if (tie_utts) {
		tie_utts->clear();
		for (int u = 0; u < n && acts; ++u)
			if (fl[2 + u] != 0) tie_utts->push_back(u);
		if (!h->use_fir) {
			g_tie_seen += (unsigned long long)n;
			g_tie_flagged += (unsigned long long)tie_utts->size();
		}
		if (tie) *tie = !tie_utts->empty();
	}
	return WC_OK;
}
// the handle that re-runs a batch after the tie flag: the same options, band-pass by direct FIR sums (created on first use)
wc_harvest *hv_exact_twin(wc_harvest *h) {
	if (h->use_fir) return h;
	if (!h->exact_twin) {
		OnDeviceOf here(h->dev);
		wc_harvest *t = wc_harvest_create(h->fs, h->f0_floor, h->f0_ceil, h->frame_period, h->target_fs, h->channels_in_octave, h->use_cos_table_opt);
		if (!t) return nullptr;
		t->use_fir = true;
		t->phases = h->phases;
		h->exact_twin = t;
	}
	return h->exact_twin;
}

// stretches [u0, u1) of consecutive utterances out of a sorted list
std::vector<std::pair<int, int>> hv_runs_of(const std::vector<int> &us) {
	std::vector<std::pair<int, int>> r;
	for (size_t k = 0; k < us.size(); ++k) {
		if (!r.empty() && r.back().second == us[k]) r.back().second = us[k] + 1;
		else r.push_back({us[k], us[k] + 1});
	}
	return r;
}

static int hv_run_device(wc_harvest *h, int n_utt, const double *d_x, const int *x_length, double *d_tpos, double *d_f0) {
	hipStream_t s = h->dev->active();
	int rc;
	for (int attempt = 0; attempt < 2; ++attempt) {
		if ((rc = hv_enqueue(h, s, n_utt, d_x, x_length, d_tpos, d_f0, attempt == 1, nullptr, nullptr, 3, nullptr, nullptr))) return rc;
		bool overflow = false, tie = false;
		std::vector<int> tied;
		if ((rc = hv_overflowed(h, s, &overflow, &tie, &tied))) return rc;
		if (overflow) continue;
		if (h->force_tie >= 0 && h->force_tie < n_utt && !h->use_fir && !h->ignore_ties && tied.empty()) tied.push_back(h->force_tie);  // (test hook)
		if (tied.empty() || !(h->phases & 1)) return WC_OK;
		// Candidates on a tie (see hv_refine_packed_kernel): the utterances that hold one -- and only those, round 6 -- once more with
		// the band-pass as direct FIR sums, stretch of consecutive utterances by stretch, straight into their places in the outputs
		// (the batch is packed: a stretch of utterances is a slice of every array).  A handle restricted to the front -- the
		// incremental stream path -- gets the twin's candidate and score rows copied into its own: the caller reads them through
		// hv_candidate_rows / hv_score_rows of THIS handle.
		wc_harvest *t = hv_exact_twin(h);
		if (!t) return WC_ERR_DEVICE;
		t->phases = h->phases;
		std::vector<long long> xo(n_utt + 1, 0), fo(n_utt + 1, 0), lo(n_utt + 1, 0);
		for (int u = 0; u < n_utt; ++u) {
			xo[u + 1] = xo[u] + x_length[u];
			fo[u + 1] = fo[u] + ((h->frame_period == 1.0) ? wc_get_samples(h->fs, x_length[u], 1) : wc_get_samples(h->fs, x_length[u], h->frame_period));
			lo[u + 1] = lo[u] + wc_get_samples(h->fs, x_length[u], 1);
		}
		for (const auto &r : hv_runs_of(tied)) {
			const int u0 = r.first, nu = r.second - r.first;
			bool done = false;
			for (int at2 = 0; at2 < 2 && !done; ++at2) {
				if ((rc = hv_enqueue(t, s, nu, d_x + xo[u0], x_length + u0, d_tpos + fo[u0], d_f0 + fo[u0], at2 == 1, nullptr, nullptr, 3, nullptr, nullptr))) return rc;
				bool o2 = false;
				if ((rc = hv_overflowed(t, s, &o2, nullptr, nullptr))) return rc;
				done = !o2;
			}
			if (!done) return fail(WC_ERR_DEVICE, "harvest: zero-crossing buffer overflow");
			if (h->phases == 1) {
				const size_t row = sizeof(double) * 7 * (size_t)h->S;
				WC_HIP(hipMemcpyAsync(h->cand1.as<char>() + row * lo[u0], t->cand1.p, row * (size_t)(lo[u0 + nu] - lo[u0]), hipMemcpyDeviceToDevice, s));
				WC_HIP(hipMemcpyAsync(h->score1.as<char>() + row * lo[u0], t->score1.p, row * (size_t)(lo[u0 + nu] - lo[u0]), hipMemcpyDeviceToDevice, s));
			}
		}
		return WC_OK;
	}
	return fail(WC_ERR_DEVICE, "harvest: zero-crossing buffer overflow");
}

wc::Device *hv_device(const wc_harvest *h) { return h->dev; }

extern "C" {

// Development hook: utterances that went through a refinement / that were flagged for a tie since the last reset (process-wide)
void wc_harvest_tie_counts(unsigned long long *seen, unsigned long long *flagged, int reset) {
	if (seen) *seen = g_tie_seen.load();
	if (flagged) *flagged = g_tie_flagged.load();
	if (reset) { g_tie_seen = 0; g_tie_flagged = 0; }
}

wc_harvest *wc_harvest_create(int fs, double f0_floor, double f0_ceil, double frame_period, double target_fs,
							  double channels_in_octave, int use_cos_table) {
	if (fs <= 0 || f0_floor <= 0 || f0_ceil <= f0_floor || frame_period <= 0 || target_fs <= 0 || channels_in_octave <= 0) {
		set_error("harvest: invalid option");
		return nullptr;
	}
	Device *dev = current_device();
	if (!dev) return nullptr;
	wc_harvest *h = new wc_harvest();
	h->fs = fs; h->f0_floor = f0_floor; h->f0_ceil = f0_ceil; h->frame_period = frame_period; h->target_fs = target_fs;
	h->channels_in_octave = channels_in_octave; h->dev = dev;
	h->use_cos_table_opt = use_cos_table;
	h->use_cos_table = use_cos_table != 0;
	// reference src/harvest.cpp:82-84, :1388-1397, :1418-1419
	h->decim = std::max(std::min(h_mround(fs / target_fs), 12), 1);
	h->fs_d = static_cast<double>(fs) / h->decim;
	const double adj_floor = f0_floor * 0.9, adj_ceil = f0_ceil * 1.1;
	h->n_bands = 1 + static_cast<int>(std::log(adj_ceil / adj_floor) / 0.69314718055994529 * channels_in_octave);
	h->max_cand = h_mround(h->n_bands / 10) * 7;
	h->S = h->max_cand / 7;
	if (h->S < 1 || h->S > MAX_SLOTS || h->n_bands < 12) {
		set_error("harvest: unsupported f0_floor / f0_ceil / channels_in_octave combination");
		delete h;
		return nullptr;
	}
	h->band_f0.resize(h->n_bands);
	h->half_len.resize(h->n_bands);
	h->tap_off.resize(h->n_bands);
	std::vector<double> taps;
	const double pi = 3.1415926535897932384;
	for (int b = 0; b < h->n_bands; ++b) {
		const double fb = adj_floor * std::pow(2.0, static_cast<double>(b + 1) / channels_in_octave);
		const int hl = h_mround(h->fs_d / fb * 2.0);  // reference :1264
		if (hl > HL_MAX) {
			set_error("harvest: f0_floor too low for the supported band-pass length (f0_floor >= 35 Hz at 8 kHz internal rate)");
			delete h;
			return nullptr;
		}
		h->band_f0[b] = fb;
		h->half_len[b] = hl;
		h->tap_off[b] = static_cast<int>(taps.size());
		const int ylen = hl * 2 + 1;
		const int nt8 = ((ylen + 7) / 8) * 8;
		for (int i = 0; i < ylen; ++i) {  // NuttallWindow * cos (reference src/world_common.cpp:118-126, src/harvest.cpp:1266-1269)
			const double t = i / (ylen - 1.0);
			double w = 0.355768 - 0.487396 * std::cos(2.0 * pi * t) + 0.144232 * std::cos(4.0 * pi * t) - 0.012604 * std::cos(6.0 * pi * t);
			w *= std::cos(2 * pi * fb * (i - hl) / h->fs_d);
			taps.push_back(w);
		}
		for (int i = ylen; i < nt8; ++i) taps.push_back(0.0);
	}
	// longest refinement window must fit the kernel's LDS buffer
	if (2 * static_cast<int>(1.5 * h->fs_d / f0_floor + 1.0) + 1 > RF_MAXW) {
		set_error("harvest: f0_floor too low for the refinement window");
		delete h;
		return nullptr;
	}
	bool ok = h->d_taps.reserve(sizeof(double) * taps.size()) == 0 && h->d_tap_off.reserve(sizeof(int) * h->n_bands) == 0 &&
			  h->d_half_len.reserve(sizeof(int) * h->n_bands) == 0 && h->d_band_f0.reserve(sizeof(double) * h->n_bands) == 0;
	ok = ok && hipMemcpy(h->d_taps.p, taps.data(), sizeof(double) * taps.size(), hipMemcpyHostToDevice) == hipSuccess;
	ok = ok && hipMemcpy(h->d_tap_off.p, h->tap_off.data(), sizeof(int) * h->n_bands, hipMemcpyHostToDevice) == hipSuccess;
	ok = ok && hipMemcpy(h->d_half_len.p, h->half_len.data(), sizeof(int) * h->n_bands, hipMemcpyHostToDevice) == hipSuccess;
	ok = ok && hipMemcpy(h->d_band_f0.p, h->band_f0.data(), sizeof(double) * h->n_bands, hipMemcpyHostToDevice) == hipSuccess;
	{
		// sliding band-pass: rotations e^{i v} for v = w, w +- Omega, w +- 2 Omega, w +- 3 Omega and P0 = e^{-i w hl}
		std::vector<double2> sd_rot(7 * h->n_bands), sd_p0(h->n_bands);
		const long double lpi = 3.14159265358979323846264338327950288L;
		for (int b = 0; b < h->n_bands; ++b) {
			const long double w = 2.0L * lpi * (long double)h->band_f0[b] / (long double)h->fs_d;
			const long double om = lpi / (long double)h->half_len[b];
			const int mult[7] = {0, 1, -1, 2, -2, 3, -3};
			for (int v = 0; v < 7; ++v) {
				const long double nu = w + mult[v] * om;
				sd_rot[7 * b + v] = make_double2((double)cosl(nu), (double)sinl(nu));
			}
			const long double ph = w * (long double)h->half_len[b];
			sd_p0[b] = make_double2((double)cosl(ph), (double)-sinl(ph));
		}
		ok = ok && h->d_sd_rot.reserve(sizeof(double2) * sd_rot.size()) == 0 && h->d_sd_p0.reserve(sizeof(double2) * sd_p0.size()) == 0 &&
			 hipMemcpy(h->d_sd_rot.p, sd_rot.data(), sizeof(double2) * sd_rot.size(), hipMemcpyHostToDevice) == hipSuccess &&
			 hipMemcpy(h->d_sd_p0.p, sd_p0.data(), sizeof(double2) * sd_p0.size(), hipMemcpyHostToDevice) == hipSuccess;
		const char *bp = getenv("WC_HARVEST_BANDPASS");
		h->use_fir = bp && std::strcmp(bp, "fir") == 0;
		const char *ti = getenv("WC_HARVEST_TIES");
		h->ignore_ties = ti && std::strcmp(ti, "ignore") == 0;
		h->force_tie = getenv("WC_HARVEST_FORCE_TIE") ? atoi(getenv("WC_HARVEST_FORCE_TIE")) : -1;
		const char *qu = getenv("WC_HARVEST_QUIET");
		h->no_quiet = qu && std::strcmp(qu, "sliding") == 0;
		const char *sl = getenv("WC_HARVEST_SDFT_LANES");
		h->sdft_lanes = sl ? atoi(sl) : 0;
		const char *sm8 = getenv("WC_HARVEST_SDFT8_MAX");
		h->sdft8_max = sm8 ? atoll(sm8) : 3072;
		h->tables_valid = false;
		h->debug_small_caps = getenv("WC_DEBUG_SMALL_CAPS") != nullptr;
		const char *dm = getenv("WC_HARVEST_DECIMATE");
		h->direct_decimation = dm && std::strcmp(dm, "direct") == 0;
		h->chunked_decimation = dm && (std::strcmp(dm, "direct") == 0 || std::strcmp(dm, "chunks") == 0);
		{
			// powers of the recursion's state matrix A = [a0 a1 a2; 1 0 0; 0 1 0] (state = the last three values of the recursion's
			// inner sequence, newest first), in extended precision: A^(C 2^j) for the scan's levels, A^(C (l + 1)) for the carry
			const DecCoef dc_ = dec_coef(h->decim);
			typedef long double LD;
			auto mul = [](const LD (&a)[9], const LD (&b)[9], LD (&o)[9]) {
				LD t[9];
				for (int i = 0; i < 3; ++i)
					for (int j = 0; j < 3; ++j) t[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
				for (int k = 0; k < 9; ++k) o[k] = t[k];
			};
			const LD A[9] = {(LD)dc_.a0, (LD)dc_.a1, (LD)dc_.a2, 1, 0, 0, 0, 1, 0};
			LD AC[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
			for (int k = 0; k < DS_C; ++k) mul(AC, A, AC);  // A^C
			LD P[9];
			for (int k = 0; k < 9; ++k) P[k] = AC[k];
			std::vector<double> ptab(64 * 9);
			for (int l = 0; l < 64; ++l) {
				for (int k = 0; k < 9; ++k) ptab[9 * l + k] = (double)P[k];
				mul(P, AC, P);
			}
			LD Q[9];
			for (int k = 0; k < 9; ++k) Q[k] = AC[k];
			for (int j = 0; j < 7; ++j) {
				for (int k = 0; k < 9; ++k) h->dec_scan.mp[j][k] = (double)Q[k];
				mul(Q, Q, Q);
			}
			ok = ok && h->d_dec_ptab.reserve(sizeof(double) * ptab.size()) == 0 &&
				 hipMemcpy(h->d_dec_ptab.p, ptab.data(), sizeof(double) * ptab.size(), hipMemcpyHostToDevice) == hipSuccess;
		}
		const char *sm = getenv("WC_HARVEST_SMOOTH");
		h->smooth_full_walk = sm && std::strcmp(sm, "full") == 0;
		const char *rfm = getenv("WC_HARVEST_REFINE");
		h->refine_mode = !rfm ? 0 : std::strcmp(rfm, "slots") == 0 ? 1 : std::strcmp(rfm, "packed") == 0 ? 2 : std::strcmp(rfm, "group2") == 0 ? 3 : std::strcmp(rfm, "group8") == 0 ? 4 : 0;
		const char *rw = getenv("WC_HARVEST_RAW");
		h->raw_from_lists = rw && std::strcmp(rw, "lists") == 0;
		h->raw_mode = rw && std::strcmp(rw, "blocks") == 0 ? 1 : rw && std::strcmp(rw, "four") == 0 ? 2 : 0;
	}
	{
		std::vector<double2> rot(2 * (RF_MAXHW + 1));
		for (int hw = 0; hw <= RF_MAXHW; ++hw) {
			const double beta = 2.0 * pi / (2 * hw + 1);
			rot[2 * hw] = make_double2(std::cos(beta), std::sin(beta));
			rot[2 * hw + 1] = make_double2(std::cos(8 * beta), std::sin(8 * beta));
		}
		ok = ok && h->d_rot.reserve(sizeof(double2) * rot.size()) == 0 &&
			 hipMemcpy(h->d_rot.p, rot.data(), sizeof(double2) * rot.size(), hipMemcpyHostToDevice) == hipSuccess;
		std::vector<double2> rot8(8 * (RF_MAXHW + 1));
		const long double lpi = 3.14159265358979323846264338327950288L;
		for (int hw = 0; hw <= RF_MAXHW; ++hw)
			for (int k = 0; k < 8; ++k) {
				const long double ang = 2.0L * lpi * k / (2 * hw + 1);
				rot8[8 * hw + k] = make_double2((double)cosl(ang), (double)sinl(ang));
			}
		ok = ok && h->d_rot8.reserve(sizeof(double2) * rot8.size()) == 0 &&
			 hipMemcpy(h->d_rot8.p, rot8.data(), sizeof(double2) * rot8.size(), hipMemcpyHostToDevice) == hipSuccess;
	}
	if (h->use_cos_table) {  // get_cos_table (reference src/harvest.cpp:152-170): one period from the first quarter, 8001 entries
		const int n = 2000;
		std::vector<double> t(n * 4 + 1, 0.0);
		const double interval = pi / 2. / n;
		for (int i = 0; i < n + 1; ++i) t[i] = std::cos(interval * i);
		for (int i = 0; i < n; ++i) t[i + n + 1] = -t[n - 1 - i];
		for (int i = 0; i < n; ++i) t[i + n * 2 + 1] = -t[i + 1];
		for (int i = 0; i < n; ++i) t[i + n * 3 + 1] = t[n - 1 - i];
		ok = ok && h->d_cos_table.reserve(sizeof(double) * t.size()) == 0 &&
			 hipMemcpy(h->d_cos_table.p, t.data(), sizeof(double) * t.size(), hipMemcpyHostToDevice) == hipSuccess;
	}
	if (!ok) {
		set_error("harvest: table upload failed");
		delete h;
		return nullptr;
	}
	dev->handle_born();
	return h;
}

int wc_harvest_get_samples(const wc_harvest *h, int x_length) {
	if (!h) return WC_ERR_INVALID;
	return wc_get_samples(h->fs, x_length, h->frame_period);
}
void wc_harvest_destroy(wc_harvest *h) {
	if (!h) return;
	h->dev->quiesce();
	if (h->exact_twin) wc_harvest_destroy(h->exact_twin);
	h->dev->handle_gone();
	for (DevBuf *b : {&h->d_cos_table, &h->d_sd_rot, &h->d_sd_p0, &h->d_slot_off, &h->d_slot_cap, &h->slots, &h->slot_count, &h->seam, &h->quiet, &h->bmax, &h->d_rot, &h->d_rot8, &h->d_dec_ptab, &h->d_taps, &h->d_tap_off, &h->d_half_len, &h->d_band_f0, &h->d_ev_band_off, &h->d_ev_cap, &h->utts, &h->dec, &h->y,
					  &h->events, &h->ev_count, &h->overflow, &h->tile_run, &h->raw, &h->cand0, &h->cand1, &h->score1, &h->cand2, &h->score2, &h->base,
					  &h->s1, &h->s2, &h->s3, &h->fixed, &h->f0_1ms, &h->sec, &h->chan, &h->smooth, &h->ibuf, &h->rawdesc, &h->d_x, &h->d_tpos, &h->d_f0})
		b->release();
	h->h_stage.release();
	h->h_x.release();
	delete h;
}

int wc_harvest_compute_device(wc_harvest *h, int n_utt, const double *d_x, const int *x_length, double *d_tpos, double *d_f0) {
	if (!h || n_utt <= 0 || !d_x || !x_length || !d_tpos || !d_f0) return fail(WC_ERR_INVALID, "harvest: null argument");
	WC_HIP(hipSetDevice(h->dev->id));
	DeviceLock lock(h->dev);
	return hv_run_device(h, n_utt, d_x, x_length, d_tpos, d_f0);
}

int wc_harvest_compute(wc_harvest *h, const double *x, int x_length, double *temporal_positions, double *f0) {
	if (!h || !x || !temporal_positions || !f0) return fail(WC_ERR_INVALID, "harvest: null argument");
	if (x_length <= 0) return fail(WC_ERR_INVALID, "harvest: bad length");
	WC_HIP(hipSetDevice(h->dev->id));
	DeviceLock lock(h->dev);
	hipStream_t s = h->dev->active();
	const int L = wc_get_samples(h->fs, x_length, h->frame_period);
	int rc;
	if ((rc = h->d_x.reserve(sizeof(double) * x_length))) return rc;
	if ((rc = h->d_tpos.reserve(sizeof(double) * L))) return rc;
	if ((rc = h->d_f0.reserve(sizeof(double) * L))) return rc;
	if ((rc = array_up(s, x, (size_t)x_length, h->h_x, h->d_x.as<double>()))) return rc;
	if ((rc = hv_run_device(h, 1, h->d_x.as<double>(), &x_length, h->d_tpos.as<double>(), h->d_f0.as<double>()))) return rc;
	WC_HIP(hipMemcpyAsync(temporal_positions, h->d_tpos.p, sizeof(double) * L, hipMemcpyDeviceToHost, s));
	WC_HIP(hipMemcpyAsync(f0, h->d_f0.p, sizeof(double) * L, hipMemcpyDeviceToHost, s));
	WC_HIP(hipStreamSynchronize(s));
	return WC_OK;
}

// Development hook (not part of include/world_class_c.h): copies an intermediate of the most recent call
// for utterance `utt` to the host.  names: y, raw, cand, score, base, fixed, f0_1ms.  Returns the number of
// doubles written (or needed when dst is NULL), <0 on error.
long long wc_harvest_debug_fetch(wc_harvest *h, const char *name, int utt, double *dst) {
	if (!h || !name || utt < 0 || utt >= (int)h->last_utts.size()) return fail(WC_ERR_INVALID, "harvest debug: bad argument");
	const HvUtt &u = h->last_utts[utt];
	const int nc = 7 * h->S;
	const double *src = nullptr;
	long long n = 0;
	std::string nm(name);
	if (nm == "y") { src = h->y.as<double>() + u.y_off; n = u.y_len; }
	else if (nm == "raw") { src = h->raw.as<double>() + u.l1_off * h->n_bands; n = (long long)u.L1 * h->n_bands; }
	else if (nm == "cand0") { src = h->cand0.as<double>() + u.l1_off * h->S; n = (long long)u.L1 * h->S; }
	else if (nm == "cand1") { src = h->cand1.as<double>() + u.l1_off * nc; n = (long long)u.L1 * nc; }
	else if (nm == "score1") { src = h->score1.as<double>() + u.l1_off * nc; n = (long long)u.L1 * nc; }
	else if (nm == "cand") { src = h->cand2.as<double>() + u.l1_off * nc; n = (long long)u.L1 * nc; }
	else if (nm == "score") { src = h->score2.as<double>() + u.l1_off * nc; n = (long long)u.L1 * nc; }
	else if (nm == "base") { src = h->base.as<double>() + u.l1_off; n = u.L1; }
	else if (nm == "s1") { src = h->s1.as<double>() + u.l1_off; n = u.L1; }
	else if (nm == "s2") { src = h->s2.as<double>() + u.l1_off; n = u.L1; }
	else if (nm == "s3") { src = h->s3.as<double>() + u.l1_off; n = u.L1; }
	else if (nm == "fixed") { src = h->fixed.as<double>() + u.l1_off; n = u.L1; }
	else if (nm == "f0_1ms") { src = h->f0_1ms.as<double>() + u.l1_off; n = u.L1; }
	else return fail(WC_ERR_INVALID, "harvest debug: unknown name");
	if (dst) {
		WC_HIP(hipMemcpyAsync(dst, src, sizeof(double) * n, hipMemcpyDeviceToHost, h->dev->active()));
		WC_HIP(hipStreamSynchronize(h->dev->active()));
	}
	return n;
}

// Development hook: the refinement kernel alone, on candidate rows given by the caller, over the utterances of the most recent
// call (their decimated signals are still in place).  cand0: [1 ms frames of the batch][S] candidate frequencies (0 = empty slot);
// cand1 / score1: [frames][7 S] refined candidates and scores.  by_slots selects the slot layout (hv_refine_kernel) instead of
// the packed one.  Lets the tests put more candidates into a frame than Harvest's detector ever finds.
int wc_harvest_debug_refine(wc_harvest *h, const double *cand0, int by_slots, double *cand1, double *score1) {
	if (!h || !cand0 || !cand1 || !score1 || !h->last_refine_valid) return fail(WC_ERR_INVALID, "harvest debug: no refinement to repeat");
	DeviceLock lock(h->dev);
	hipStream_t s = h->dev->active();
	const RefArgs fa = h->last_refine;
	const size_t n_in = (size_t)fa.total_frames * fa.p.S, n_out = (size_t)fa.total_frames * fa.p.n_cand;
	WC_HIP(hipMemcpyAsync(const_cast<double *>(fa.cand0), cand0, sizeof(double) * n_in, hipMemcpyHostToDevice, s));
	launch_refine(fa, s, by_slots == 1 ? 1 : by_slots == 2 ? 2 : 0, h->use_cos_table);  // (by_slots: 0 default, 1 slots, 2 packed)
	WC_HIP(hipGetLastError());
	WC_HIP(hipMemcpyAsync(cand1, fa.cand1, sizeof(double) * n_out, hipMemcpyDeviceToHost, s));
	WC_HIP(hipMemcpyAsync(score1, fa.score1, sizeof(double) * n_out, hipMemcpyDeviceToHost, s));
	WC_HIP(hipStreamSynchronize(s));
	return WC_OK;
}

}  // extern "C"
