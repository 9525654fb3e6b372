// Does a D2H copy keep the link's idle rate while full-grid FP64 kernels run, if those kernels leave a few CUs alone?
// (development aid, round 5: the host front-end's rows leave at 42-50 GB/s under load against 57 GB/s on an idle chip;
// nearly all of its copies are done by blit kernels, which wait for wavefront places like everything else)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/cumask_probe tools/cumask_probe.hip && /tmp/cumask_probe
// Stream A: a burn kernel shaped like the 48 kHz wavefront kernels (two wavefronts per SIMD, 64-thread workgroups, ~60 us each),
// on a plain stream or on one created with hipExtStreamCreateWithCUMask that leaves R CUs of every XCD out.
// Streams B1 / B2 (high priority): 16 MB D2H copies into page-locked memory, 512 MB in all, two in flight.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void burn(double *out, int iters) {
	double d[32];
#pragma unroll
	for (int k = 0; k < 32; ++k) d[k] = threadIdx.x * 1e-3 + k;
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int k = 0; k < 32; ++k) d[k] = fma(d[k], 1.0000001, 1e-9);
	}
	double s = 0;
#pragma unroll
	for (int k = 0; k < 32; ++k) s += d[k];
	if (s == 123.456) out[blockIdx.x] = s;
}

// a kernel that does nothing but move memory: what the D4C / Synthesis kernels' parked rows do to the fabric, without their arithmetic
__global__ __launch_bounds__(256) void stream_rw(const double2 *__restrict__ src, double2 *__restrict__ dst, long long n) {
	for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += 256ll * gridDim.x) dst[i] = src[i];
}

int main() {
	setvbuf(stdout, nullptr, _IONBF, 0);
	hipDeviceProp_t prop;
	CK(hipGetDeviceProperties(&prop, 0));
	const int n_cu = prop.multiProcessorCount;
	std::printf("%s, %d CUs\n", prop.name, n_cu);
	const size_t total = 512ull << 20, piece = 16ull << 20;
	char *d_src, *h_dst;
	double *d_out;
	CK(hipMalloc(&d_src, total));
	CK(hipMemset(d_src, 1, total));
	CK(hipHostMalloc(&h_dst, total, hipHostMallocDefault));
	CK(hipMalloc(&d_out, sizeof(double) * (1 << 22)));
	int lo, hi;
	CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
	hipStream_t b1, b2;
	CK(hipStreamCreateWithPriority(&b1, hipStreamNonBlocking, hi));
	CK(hipStreamCreateWithPriority(&b2, hipStreamNonBlocking, hi));
	auto copies = [&]() -> double {
		auto t0 = std::chrono::steady_clock::now();
		int k = 0;
		for (size_t o = 0; o < total; o += piece, ++k) (void)hipMemcpyAsync(h_dst + o, d_src + o, piece, hipMemcpyDeviceToHost, (k & 1) ? b2 : b1);
		(void)hipStreamSynchronize(b1);
		(void)hipStreamSynchronize(b2);
		return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	};
	copies();
	std::printf("idle chip:                       %.1f GB/s\n", total / copies() / 1e9);
	// masks: bit c of the mask = CU c; the driver deals CUs to XCDs round robin (CU c on XCD c % 8), so leaving out the CUs
	// 8 r .. 8 r + 7 leaves one CU of every XCD out
	for (int reserve : {0}) {  // (1, 2: 54.8 GB/s as well, the burns 5 - 10 % slower; 4, and whatever followed the masked streams in the process: never returned on this ROCm)
		std::vector<uint32_t> mask((n_cu + 31) / 32, 0xFFFFFFFFu);
		for (int c = 0; c < 8 * reserve && c < n_cu; ++c) mask[c / 32] &= ~(1u << (c % 32));
		hipStream_t a;
		if (reserve == 0) CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
		else CK(hipExtStreamCreateWithCUMask(&a, (uint32_t)mask.size(), mask.data()));
		// how long the burn takes alone
		const int blocks = 2048 * 40, iters = 600;
		hipEvent_t e0, e1;
		CK(hipEventCreate(&e0));
		CK(hipEventCreate(&e1));
		hipLaunchKernelGGL(burn, dim3(blocks), dim3(64), 0, a, d_out, iters);
		CK(hipStreamSynchronize(a));
		CK(hipEventRecord(e0, a));
		hipLaunchKernelGGL(burn, dim3(blocks), dim3(64), 0, a, d_out, iters);
		CK(hipEventRecord(e1, a));
		CK(hipStreamSynchronize(a));
		float alone = 0;
		CK(hipEventElapsedTime(&alone, e0, e1));
		// copies beside a train of burns
		CK(hipEventRecord(e0, a));
		for (int r = 0; r < 6; ++r) hipLaunchKernelGGL(burn, dim3(blocks), dim3(64), 0, a, d_out, iters);
		CK(hipEventRecord(e1, a));
		const double t = copies();
		CK(hipStreamSynchronize(a));
		float six = 0;
		CK(hipEventElapsedTime(&six, e0, e1));
		std::printf("kernels leave %d CU(s) per XCD out: burn alone %.2f ms; copies beside six burns %.1f GB/s (%.1f ms for the copies, %.1f ms for the burns)\n",
					reserve, alone, total / t / 1e9, t * 1e3, six);
		CK(hipStreamDestroy(a));
	}
	{
		// copies beside a memory-bound kernel (1 GB read + 1 GB written per launch)
		const long long n = (1ll << 30) / 16;
		double2 *ma, *mb;
		CK(hipMalloc(&ma, n * 16));
		CK(hipMalloc(&mb, n * 16));
		CK(hipMemset(ma, 0, n * 16));
		hipStream_t a;
		CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
		hipEvent_t e0, e1;
		CK(hipEventCreate(&e0));
		CK(hipEventCreate(&e1));
		hipLaunchKernelGGL(stream_rw, dim3(4096), dim3(256), 0, a, ma, mb, n);
		CK(hipStreamSynchronize(a));
		CK(hipEventRecord(e0, a));
		for (int r = 0; r < 40; ++r) hipLaunchKernelGGL(stream_rw, dim3(4096), dim3(256), 0, a, ma, mb, n);
		CK(hipEventRecord(e1, a));
		const double t = copies();
		CK(hipStreamSynchronize(a));
		float ms = 0;
		CK(hipEventElapsedTime(&ms, e0, e1));
		std::printf("copies beside a memory-bound kernel (%.2f TB/s of HBM traffic for %.1f ms): %.1f GB/s (%.1f ms for the copies)\n",
					40 * 2.0 * (double)(n * 16) / (ms * 1e-3) / 1e12, ms, total / t / 1e9, t * 1e3);
	}
	return 0;
}
