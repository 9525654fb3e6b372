// How much instruction-level parallelism does ONE wavefront need to keep its FP64 issue rate?  (development aid)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/chain tools/fp64_chain_latency.hip && /tmp/chain
// Every lane runs K independent chains of v_fma_f64 (K = 1, 2, 4, 8: the next instruction of a chain needs the previous one's
// result), 8 instructions per trip, at 1, 2, 3 and 4 wavefronts per SIMD.  ns per instruction and SIMD from HIP events.
#include <hip/hip_runtime.h>

#include <cstdio>

template <int K>
__global__ __launch_bounds__(256) void burn(double *out, int iters) {
	double d[8];
#pragma unroll
	for (int k = 0; k < 8; ++k) d[k] = threadIdx.x * 1e-3 + k;
	const double m = 1.0000001, c = 1e-9;
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int r = 0; r < 8 / K; ++r)
#pragma unroll
			for (int k = 0; k < K; ++k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[k]) : "v"(m), "v"(c));
	}
	double s = 0;
#pragma unroll
	for (int k = 0; k < 8; ++k) s += d[k];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int K>
void run(int waves, double *out) {
	const int iters = 1 << 15;
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	const int blocks = 256 * waves;  // 256 CUs x `waves` workgroups of four wavefronts: `waves` wavefronts per SIMD
	burn<K><<<blocks, 256>>>(out, 16);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	burn<K><<<blocks, 256>>>(out, iters);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms;
	hipEventElapsedTime(&ms, e0, e1);
	const double per_simd = (double)iters * 8 * waves;
	printf("  %d chain%s, %d wavefront%s per SIMD: %.3f ns per instruction and SIMD\n", K, K > 1 ? "s" : " ", waves, waves > 1 ? "s" : " ", ms * 1e6 / per_simd);
}

int main() {
	double *out;
	hipMalloc(&out, sizeof(double) * 256 * 256 * 8);
	for (int w = 1; w <= 4; ++w) {
		run<1>(w, out); run<2>(w, out); run<4>(w, out); run<8>(w, out);
	}
	return 0;
}
