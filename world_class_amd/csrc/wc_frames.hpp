// Helpers shared by the per-frame stage kernels: utterance lookup, XCD-aware frame order, RNG table
// access, per-utterance stream-offset scan.
#pragma once
#include "wc_device.hpp"
#include "wc_internal.hpp"

namespace wc {

__device__ __forceinline__ double randn_at(const uint32_t *__restrict__ table, unsigned long long idx) {
	return table[idx] / 268435456.0 - 6.0;
}
__device__ __forceinline__ int find_utt(const UttDesc *__restrict__ utts, int n_utt, long long frame) {
	int lo = 0, hi = n_utt - 1;
	while (lo < hi) {
		int mid = (lo + hi + 1) >> 1;
		if (utts[mid].f_off <= frame) lo = mid; else hi = mid - 1;
	}
	return lo;
}
// blockIdx -> frame so that each XCD (block b runs on XCD b % 8) walks one contiguous range of
// frames: neighbouring frames share almost all of their input samples, which then stay in that
// XCD's L2.
__device__ __forceinline__ long long xcd_frame(long long b, long long total) {
	long long per = (total + 7) / 8;
	return (b & 7) * per + (b >> 3);
}

// one block per utterance: off[frame] = start + exclusive prefix of cnt over the utterance's frames,
// end_pos[u] = start + total; start = start[u] when given, else utt.rng_pos
static __global__ void utt_scan_kernel(const uint32_t *__restrict__ cnt, const UttDesc *__restrict__ utts,
								const unsigned long long *__restrict__ start, unsigned long long *__restrict__ off, unsigned long long *__restrict__ end_pos) {
	__shared__ unsigned long long s[256];
	const UttDesc u = utts[blockIdx.x];
	unsigned long long carry = start ? start[blockIdx.x] : u.rng_pos;
	int tid = threadIdx.x;
	for (int base = 0; base < u.f_len; base += 256) {
		int i = base + tid;
		unsigned long long v = (i < u.f_len) ? cnt[u.f_off + i] : 0ull;
		s[tid] = v;
		__syncthreads();
		for (int o = 1; o < 256; o <<= 1) {
			unsigned long long t = (tid >= o) ? s[tid - o] : 0ull;
			__syncthreads();
			s[tid] += t;
			__syncthreads();
		}
		if (i < u.f_len) off[u.f_off + i] = carry + s[tid] - v;
		carry += s[255];
		__syncthreads();
	}
	if (tid == 0) end_pos[blockIdx.x] = carry;
}


}  // namespace wc
