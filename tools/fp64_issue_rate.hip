// What FP64 vector issue rate does this MI355X sustain?  (development aid; numbers quoted in DESIGN.md section 5)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fp64_issue_rate tools/fp64_issue_rate.hip && /tmp/fp64_issue_rate
// Every lane runs 8 independent FP64 FMA chains; the launch sizes put 1/4, 1, 2, 4 and 8 wavefronts on every SIMD.
// Nominal peak: 256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles = 614.4 G wave-instructions/s (78.6 TFLOP/s).
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ void burn(double *out, long long *clk, int iters, int mode) {
	double a0 = threadIdx.x * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
	const double m = 1.0000001, c = 1e-9;
	const long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
	for (int i = 0; i < iters; ++i) {
		if (mode == 0) {
			a0 = fma(a0, m, c); a1 = fma(a1, m, c); a2 = fma(a2, m, c); a3 = fma(a3, m, c);
			a4 = fma(a4, m, c); a5 = fma(a5, m, c); a6 = fma(a6, m, c); a7 = fma(a7, m, c);
		} else {
			asm volatile("s_sleep 8");  // (unused mode: idle loop)
		}
	}
	const long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
	out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
	if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
	const int threads = 256;
	double *out;
	long long *clk;
	CK(hipMalloc(&out, sizeof(double) * 256 * 8 * threads));
	CK(hipMalloc(&clk, sizeof(long long) * 2 * 256 * 8));
	// waves per SIMD: 1/4 (one wave on one SIMD of every CU), 1, 2, 4, 8
	const int cfg[5][2] = {{256, 64}, {256, 256}, {512, 256}, {1024, 256}, {2048, 256}};
	for (int c = 0; c < 5; ++c)
		for (int rep = 0; rep < 2; ++rep) {
			const int blocks = cfg[c][0], th = cfg[c][1], iters = 1000000;
			hipEvent_t e0, e1;
			CK(hipEventCreate(&e0));
			CK(hipEventCreate(&e1));
			CK(hipEventRecord(e0));
			hipLaunchKernelGGL(burn, dim3(blocks), dim3(th), 0, 0, out, clk, iters, 0);
			CK(hipEventRecord(e1));
			CK(hipEventSynchronize(e1));
			float ms;
			CK(hipEventElapsedTime(&ms, e0, e1));
			const double waves = (double)blocks * (th / 64);
			const double rate = waves * 8.0 * iters / (ms * 1e-3) / 1e9;  // G wave-FMA / s
			// one FP64 wave instruction occupies its SIMD for 4 cycles: a SIMD that is never starved issues clock / 4 per second
			const double busy_simds = waves < 1024 ? waves : 1024;
			std::printf("%5d waves (%4.2f per SIMD): %7.1f ms, %6.1f G wave-FMA/s = %5.1f TFLOP/s f64; per busy SIMD %.0f M/s -> implied clock %.0f MHz\n",
						(int)waves, waves / 1024.0, ms, rate, rate * 128 / 1e3, rate * 1e3 / busy_simds, rate * 1e3 / busy_simds * 4);
		}
	return 0;
}
