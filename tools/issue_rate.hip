// What does a vector instruction of each class cost on this MI355X?  (development aid; the numbers behind bench.py's
// `issue_cycles_frac` and DESIGN.md section 3; successor of tools/fp64_issue_rate.hip, which timed one FP64 FMA stream)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/issue_rate tools/issue_rate.hip && /tmp/issue_rate [out.json]
// Every lane runs 8 independent chains of ONE instruction (inline asm, so the class is what is measured), 16384 x 8 instructions
// per timed stretch, at 1, 2, 3, 4 and 8 wavefronts per SIMD (256-thread workgroups, 256 x W of them).  Lane 0 of every wavefront
// reads the shader clock (s_memtime) and the 100 MHz wall clock around the stretch:
//   cycles per instruction and SIMD = shader cycles of the stretch / (instructions of one wavefront x wavefronts on its SIMD)
//   clock = shader cycles / wall time  (the chip clocks to its power budget: an FP64 stream runs well below 2.4 GHz)
// "mixed" streams alternate two classes: if their cost is the sum of the parts the classes share one issue port.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

enum Mode { FMA64, ADD64, MUL64, FMA32, ADD_U32, LSHL_ADD_U64, CNDMASK, MOV_B32, MOV_B64, CVT_F64_I32, CVT_I32_F64, CMP_F64, MIX_FMA64_ADDU32, MIX_FMA64_CNDMASK,
			MIX_FMA64_MOV64, N_MODES };
static const char *kNames[N_MODES] = {"v_fma_f64", "v_add_f64", "v_mul_f64", "v_fma_f32", "v_add_u32", "v_lshl_add_u64", "v_cndmask_b32", "v_mov_b32",
									   "v_mov_b64", "v_cvt_f64_i32", "v_cvt_i32_f64", "v_cmp_gt_f64", "mixed v_fma_f64 + v_add_u32", "mixed v_fma_f64 + v_cndmask_b32",
									   "mixed v_fma_f64 + v_mov_b64"};
// instructions per unrolled group of 8 chains (mixed streams issue two per chain)
static const int kPerChain[N_MODES] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2};

template <int MODE>
__global__ __launch_bounds__(256) void burn(double *out, long long *clk, int iters) {
	double d[8], e[8];
	int i32[8], j32[8];
	float f[8];
	unsigned long long u[8];
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		d[k] = threadIdx.x * 1e-3 + k; e[k] = 1.0 + k; i32[k] = threadIdx.x + k; j32[k] = k; f[k] = k + 0.5f; u[k] = threadIdx.x + k;
	}
	const double m = 1.0000001, c = 1e-9;
	const float mf = 1.0001f, cf = 1e-6f;
	const int one = 1;
	asm volatile("v_cmp_gt_i32 vcc, %0, %1" ::"v"(i32[0]), "v"(32) : "vcc");
	const long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
	for (int it = 0; it < iters; ++it) {
#define FMA64_(k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[k]) : "v"(m), "v"(c));
#define ADD64_(k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[k]) : "v"(c));
#define MUL64_(k) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[k]) : "v"(m));
#define FMA32_(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[k]) : "v"(mf), "v"(cf));
#define ADDU_(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(i32[k]) : "v"(one));
#define LSHL_(k) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));
#define CND_(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(j32[k]) : "v"(one) : "vcc");
#define MOV32_(k) asm volatile("v_mov_b32 %0, %1" : "=v"(j32[k]) : "v"(i32[k]));
#define MOV64_(k) asm volatile("v_mov_b64 %0, %1" : "=v"(e[k]) : "v"(d[k]));
#define CVTDI_(k) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(e[k]) : "v"(i32[k]));
#define CVTID_(k) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(j32[k]) : "v"(d[k]));
#define CMP64_(k) asm volatile("v_cmp_gt_f64 vcc, %0, %1" ::"v"(d[k]), "v"(e[k]) : "vcc");
		if (MODE == FMA64) { REP8(FMA64_) }
		if (MODE == ADD64) { REP8(ADD64_) }
		if (MODE == MUL64) { REP8(MUL64_) }
		if (MODE == FMA32) { REP8(FMA32_) }
		if (MODE == ADD_U32) { REP8(ADDU_) }
		if (MODE == LSHL_ADD_U64) { REP8(LSHL_) }
		if (MODE == CNDMASK) { REP8(CND_) }
		if (MODE == MOV_B32) { REP8(MOV32_) }
		if (MODE == MOV_B64) { REP8(MOV64_) }
		if (MODE == CVT_F64_I32) { REP8(CVTDI_) }
		if (MODE == CVT_I32_F64) { REP8(CVTID_) }
		if (MODE == CMP_F64) { REP8(CMP64_) }
#define MIXA_(k) FMA64_(k) ADDU_(k)
#define MIXB_(k) FMA64_(k) CND_(k)
#define MIXC_(k) FMA64_(k) asm volatile("v_mov_b64 %0, %1" : "=v"(e[k]) : "v"(u[k]));
		if (MODE == MIX_FMA64_ADDU32) { REP8(MIXA_) }
		if (MODE == MIX_FMA64_CNDMASK) { REP8(MIXB_) }
		if (MODE == MIX_FMA64_MOV64) { REP8(MIXC_) }
	}
	const long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
	double s = 0;
#pragma unroll
	for (int k = 0; k < 8; ++k) s += d[k] + e[k] + i32[k] + j32[k] + f[k] + (double)u[k];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
	if ((threadIdx.x & 63) == 0) {
		const int wv = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
		clk[2 * wv] = t1 - t0;
		clk[2 * wv + 1] = w1 - w0;
	}
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef void (*KernelFn)(double *, long long *, int);

int main(int argc, char **argv) {
	const int threads = 256, iters = 16384;  // (1 - 5 ms per launch at 8 wavefronts per SIMD: the length of the product's kernels)
	const int max_blocks = 256 * 8;
	double *out;
	long long *clk;
	CK(hipMalloc(&out, sizeof(double) * max_blocks * threads));
	CK(hipMalloc(&clk, sizeof(long long) * 2 * max_blocks * 4));
	KernelFn fn[N_MODES] = {burn<FMA64>, burn<ADD64>, burn<MUL64>, burn<FMA32>, burn<ADD_U32>, burn<LSHL_ADD_U64>, burn<CNDMASK>, burn<MOV_B32>, burn<MOV_B64>,
							burn<CVT_F64_I32>, burn<CVT_I32_F64>, burn<CMP_F64>, burn<MIX_FMA64_ADDU32>, burn<MIX_FMA64_CNDMASK>, burn<MIX_FMA64_MOV64>};
	std::string json = "{\n";
	std::printf("%-32s %5s  %9s %9s %9s %9s %9s   clock at 8 waves\n", "stream", "", "1 w/SIMD", "2", "3", "4", "8");
	const int Ws[5] = {1, 2, 3, 4, 8};
	for (int md = 0; md < N_MODES; ++md) {
		double cpi[5] = {0, 0, 0, 0, 0}, mhz[5] = {0, 0, 0, 0, 0}, nspi[5] = {0, 0, 0, 0, 0};
		hipEvent_t e0, e1;
		CK(hipEventCreate(&e0));
		CK(hipEventCreate(&e1));
		for (int wi = 0; wi < 5; ++wi) {
			const int W = Ws[wi], blocks = 256 * W, waves = blocks * 4;
			std::vector<long long> h(2 * (size_t)waves);
			float ms = 0;
			for (int rep = 0; rep < 2; ++rep) {  // (the second run is the one that counts: clocks have settled)
				CK(hipEventRecord(e0));
				hipLaunchKernelGGL(fn[md], dim3(blocks), dim3(threads), 0, 0, out, clk, iters);
				CK(hipEventRecord(e1));
				CK(hipDeviceSynchronize());
				CK(hipEventElapsedTime(&ms, e0, e1));
			}
			// the clock-free figure: nanoseconds of one SIMD per instruction, from the launch's duration on the host's events
			nspi[wi] = (double)ms * 1e6 / ((double)iters * 8 * kPerChain[md] * W);
			CK(hipMemcpy(h.data(), clk, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
			double cyc = 0, wall = 0;
			for (int v = 0; v < waves; ++v) { cyc += (double)h[2 * v]; wall += (double)h[2 * v + 1]; }
			cyc /= waves; wall /= waves;
			cpi[wi] = cyc / ((double)iters * 8 * kPerChain[md] * W);
			mhz[wi] = cyc / wall * 100.0;
		}
		std::printf("%-32s cyc/inst/SIMD %9.2f %9.2f %9.2f %9.2f %9.2f   %6.0f MHz\n", kNames[md], cpi[0], cpi[1], cpi[2], cpi[3], cpi[4], mhz[4]);
		std::printf("%-32s  ns/inst/SIMD %9.3f %9.3f %9.3f %9.3f %9.3f   (HIP events around the launch)\n", "", nspi[0], nspi[1], nspi[2], nspi[3], nspi[4]);
		char buf[1024];
		std::snprintf(buf, sizeof buf, "  \"%s\": {\"ns_per_inst_by_waves\": {\"1\": %.4f, \"2\": %.4f, \"3\": %.4f, \"4\": %.4f, \"8\": %.4f}, \"cycles_per_inst_by_waves\": {\"1\": %.3f, \"2\": %.3f, \"3\": %.3f, \"4\": %.3f, \"8\": %.3f}, \"clock_mhz_at_8\": %.0f, \"clock_mhz_at_2\": %.0f}%s\n",
					  kNames[md], nspi[0], nspi[1], nspi[2], nspi[3], nspi[4], cpi[0], cpi[1], cpi[2], cpi[3], cpi[4], mhz[4], mhz[1], md + 1 < N_MODES ? "," : "");
		json += buf;
	}
	json += "}\n";
	if (argc > 1) {
		FILE *f = std::fopen(argv[1], "w");
		if (f) { std::fputs(json.c_str(), f); std::fclose(f); }
	}
	return 0;
}
