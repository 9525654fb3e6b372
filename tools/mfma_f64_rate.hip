// What does an FP64 matrix instruction cost on this MI355X, alone and beside a vector FP64 stream?  (development aid; the numbers
// behind DESIGN.md section 3's answer to the round-5 verdict's item 7: could the constant-matrix stages of the wavefront FFT -- a
// radix-16 butterfly is a constant 16 x 16 complex matrix applied to 64 columns -- run on the matrix pipe beside the VALU?)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rate tools/mfma_f64_rate.hip && /tmp/mfma_rate
// Streams, every wavefront the same, 4096 trips, at 1 / 2 / 4 wavefronts per SIMD:
//   mfma16      8 independent v_mfma_f64_16x16x4_f64 per trip (2048 FLOP each)
//   mfma4       8 independent v_mfma_f64_4x4x4_4b_f64 per trip (512 FLOP each)
//   fma         32 independent v_fma_f64 per trip (128 FLOP each)
//   mfma16+fma  the two interleaved, 8 + 32 per trip: if the pipes run side by side it costs max(the two), if not their sum
#include <hip/hip_runtime.h>

#include <cstdio>

typedef double double4_ __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void burn(double *out, int iters) {
	double4_ acc[8];
	double acc1[8];
	double d[32];
	const double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9;
#pragma unroll
	for (int k = 0; k < 8; ++k) { acc[k] = double4_{0.0, 0.0, 0.0, 0.0}; acc1[k] = 0.0; }
#pragma unroll
	for (int k = 0; k < 32; ++k) d[k] = k + threadIdx.x * 1e-3;
	const double m = 1.0000001, c = 1e-9;
	for (int it = 0; it < iters; ++it) {
		if (MODE == 0 || MODE == 3) {
#pragma unroll
			for (int k = 0; k < 8; ++k) {
				acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k], 0, 0, 0);
				if (MODE == 3) {
#pragma unroll
					for (int j = 0; j < 4; ++j) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[4 * k + j]) : "v"(m), "v"(c));
				}
			}
		}
		if (MODE == 1) {
#pragma unroll
			for (int k = 0; k < 8; ++k) acc1[k] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc1[k], 0, 0, 0);
		}
		if (MODE == 2) {
#pragma unroll
			for (int k = 0; k < 32; ++k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[k]) : "v"(m), "v"(c));
		}
	}
	double s = 0;
#pragma unroll
	for (int k = 0; k < 8; ++k) s += acc[k].x + acc[k].y + acc[k].z + acc[k].w + acc1[k];
#pragma unroll
	for (int k = 0; k < 32; ++k) s += d[k];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef void (*KernelFn)(double *, int);

int main() {
	const int iters = 4096;
	double *out;
	CK(hipMalloc(&out, sizeof(double) * 256 * 4 * 256));
	KernelFn fn[4] = {burn<0>, burn<1>, burn<2>, burn<3>};
	const char *names[4] = {"mfma16 (8 per trip)", "mfma4 (8 per trip)", "fma (32 per trip)", "mfma16 + fma (8 + 32 per trip)"};
	const double flop_per_trip[4] = {8 * 2048.0, 8 * 512.0, 32 * 128.0, 8 * 2048.0 + 32 * 128.0};
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	std::printf("%-34s %5s %14s %14s %14s\n", "stream", "", "1 wave / SIMD", "2", "4");
	for (int md = 0; md < 4; ++md) {
		double ns[3], tf[3];
		const int Ws[3] = {1, 2, 4};
		for (int wi = 0; wi < 3; ++wi) {
			const int blocks = 256 * Ws[wi];
			float ms = 0;
			for (int rep = 0; rep < 2; ++rep) {
				CK(hipEventRecord(e0));
				hipLaunchKernelGGL(fn[md], dim3(blocks), dim3(256), 0, 0, out, iters);
				CK(hipEventRecord(e1));
				CK(hipDeviceSynchronize());
				CK(hipEventElapsedTime(&ms, e0, e1));
			}
			ns[wi] = (double)ms * 1e6 / ((double)iters * Ws[wi]);  // nanoseconds of one SIMD per trip of one wavefront
			tf[wi] = flop_per_trip[md] * iters * (double)blocks * 4 / (ms * 1e-3) / 1e12;
		}
		std::printf("%-34s ns / trip / SIMD %10.1f %14.1f %14.1f\n", names[md], ns[0], ns[1], ns[2]);
		std::printf("%-34s TFLOP/s (chip)   %10.1f %14.1f %14.1f\n", "", tf[0], tf[1], tf[2]);
	}
	return 0;
}
