// Library-wide plumbing of the C-ABI: error state, per-device state (stream, twiddle table, RNG draw
// table), device buffers, size helpers.  No signal-path arithmetic lives here except the RNG table.
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>

#include "wc_device.hpp"
#include "wc_internal.hpp"
#include "wc_wavefft.hpp"

namespace wc {

static thread_local std::string g_error;
void set_error(const std::string &msg) { g_error = msg; }
int fail(int code, const std::string &msg) {
	g_error = msg;
	return code;
}

int DevBuf::reserve(size_t bytes) {
	if (bytes <= cap) return WC_OK;
	if (p) (void)hipFree(p);
	p = nullptr;
	cap = 0;
	size_t want = bytes + bytes / 4 + 256;
	WC_HIP(hipMalloc(&p, want));
	cap = want;
	return WC_OK;
}
void DevBuf::release() {
	if (p) (void)hipFree(p);
	p = nullptr;
	cap = 0;
}
int HostBuf::mark(hipStream_t s) {
	if (!ev) WC_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
	WC_HIP(hipEventRecord(ev, s));
	return WC_OK;
}
int HostBuf::reserve(size_t bytes) {
	if (ev) WC_HIP(hipEventSynchronize(ev));  // the previous call's copies have read the buffer
	if (bytes <= cap) return WC_OK;
	if (p) (void)hipHostFree(p);
	p = nullptr;
	cap = 0;
	size_t want = bytes + bytes / 4 + 256;
	WC_HIP(hipHostMalloc(&p, want, hipHostMallocDefault));
	cap = want;
	return WC_OK;
}
void HostBuf::release() {
	if (ev) { (void)hipEventSynchronize(ev); (void)hipEventDestroy(ev); ev = nullptr; }
	if (p) (void)hipHostFree(p);
	p = nullptr;
	cap = 0;
}

// ------------------------------------------------------------------------------------------------
// RNG: xorshift128 with the reference's draw structure (reference
// src/world_matlabfunctions.cpp:243-264): per draw one shift-only step (w not updated), then 12
// full steps; the draw is the sum of (w >> 4) over the 12 steps.  The state transition of one draw
// is linear over GF(2), so stream position p is reachable by multiplying the seed state with the
// binary matrix T^p (host side, below); the device then fills the table in independent chunks.
// ------------------------------------------------------------------------------------------------
struct XorShift {
	uint32_t x, y, z, w;
	__host__ __device__ uint32_t draw() {
		uint32_t t = x ^ (x << 11);
		x = y; y = z; z = w;
		uint32_t acc = 0;
#pragma unroll
		for (int i = 0; i < 12; ++i) {
			t = x ^ (x << 11);
			x = y; y = z; z = w;
			w = (w ^ (w >> 19)) ^ (t ^ (t >> 8));
			acc += w >> 4;
		}
		return acc;
	}
};

struct BitMat { uint32_t col[128][4]; };
static void matvec(const BitMat &m, const uint32_t s[4], uint32_t o[4]) {
	o[0] = o[1] = o[2] = o[3] = 0;
	for (int j = 0; j < 128; ++j)
		if ((s[j >> 5] >> (j & 31)) & 1u)
			for (int k = 0; k < 4; ++k) o[k] ^= m.col[j][k];
}
static std::vector<BitMat> build_jump() {
	std::vector<BitMat> tab(64);
	for (int j = 0; j < 128; ++j) {
		uint32_t s[4] = {0, 0, 0, 0};
		s[j >> 5] = 1u << (j & 31);
		XorShift r{s[0], s[1], s[2], s[3]};
		r.draw();
		tab[0].col[j][0] = r.x; tab[0].col[j][1] = r.y; tab[0].col[j][2] = r.z; tab[0].col[j][3] = r.w;
	}
	for (int k = 1; k < 64; ++k)
		for (int j = 0; j < 128; ++j) matvec(tab[k - 1], tab[k - 1].col[j], tab[k].col[j]);
	return tab;
}
static const std::vector<BitMat> &jump() {
	static const std::vector<BitMat> tab = build_jump();
	return tab;
}
void rng_state_at(uint64_t position, uint32_t s[4]) {
	s[0] = 123456789u; s[1] = 362436069u; s[2] = 521288629u; s[3] = 88675123u;
	const std::vector<BitMat> &tab = jump();
	uint32_t o[4];
	for (int k = 0; k < 64; ++k)
		if ((position >> k) & 1ull) { matvec(tab[k], s, o); std::memcpy(s, o, sizeof(o)); }
}

constexpr int kRngChunk = 256;  // draws per device thread
__global__ void rng_fill_kernel(const uint4 *__restrict__ seeds, uint32_t *__restrict__ table, int n_chunks) {
	int c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= n_chunks) return;
	uint4 s = seeds[c];
	XorShift r{s.x, s.y, s.z, s.w};
	uint4 *out = reinterpret_cast<uint4 *>(table + (size_t)c * kRngChunk);
	for (int i = 0; i < kRngChunk / 4; ++i) {
		uint4 v;
		v.x = r.draw(); v.y = r.draw(); v.z = r.draw(); v.w = r.draw();
		out[i] = v;
	}
}

// The chunk seeds T^(256 c) s0 on the device: the host jumps from super-chunk to super-chunk (kRngSuper chunks each, one 128 x 128
// bit-matrix product per jump) and every thread here walks its super-chunk's chunks with T^256 -- on the host alone the 124 k
// products behind the table of one 48 kHz x 10 s D4C call were 35 of the 45 ms of a process's first compute().
constexpr int kRngSuper = 64;
__global__ void rng_seed_expand_kernel(const uint4 *__restrict__ super, const uint4 *__restrict__ step, uint4 *__restrict__ seeds, int n_chunks) {
	__shared__ uint4 m[128];  // column j of T^256
	for (int j = threadIdx.x; j < 128; j += blockDim.x) m[j] = step[j];
	__syncthreads();
	const int t = blockIdx.x * blockDim.x + threadIdx.x;
	if ((long long)t * kRngSuper >= n_chunks) return;
	uint4 s = super[t];
	for (int c = 0; c < kRngSuper; ++c) {
		const int at = t * kRngSuper + c;
		if (at >= n_chunks) break;
		seeds[at] = s;
		uint4 o = make_uint4(0u, 0u, 0u, 0u);
		const uint32_t w[4] = {s.x, s.y, s.z, s.w};
#pragma unroll 4
		for (int j = 0; j < 128; ++j) {
			const uint32_t on = 0u - ((w[j >> 5] >> (j & 31)) & 1u);
			const uint4 col = m[j];
			o.x ^= col.x & on; o.y ^= col.y & on; o.z ^= col.z & on; o.w ^= col.w & on;
		}
		s = o;
	}
}

int launch_rng_fill(Device *dev, uint32_t *table, uint64_t first, uint64_t count) {
	// first and count are multiples of kRngChunk
	int n_chunks = (int)(count / kRngChunk);
	if (n_chunks == 0) return WC_OK;
	const int n_super = (n_chunks + kRngSuper - 1) / kRngSuper;
	std::vector<uint32_t> host((size_t)n_super * 4 + 128 * 4);
	uint32_t s[4], o[4];
	rng_state_at(first, s);
	const BitMat &jump_super = jump()[14];  // T^(256 * 64)
	static_assert(kRngChunk == 256 && kRngSuper == 64, "jump()[8] is T^256, jump()[14] is T^(256 * 64)");
	for (int c = 0; c < n_super; ++c) {
		std::memcpy(&host[(size_t)c * 4], s, sizeof(s));
		matvec(jump_super, s, o);
		std::memcpy(s, o, sizeof(o));
	}
	std::memcpy(&host[(size_t)n_super * 4], jump()[8].col, sizeof(uint32_t) * 128 * 4);
	uint4 *d_host = nullptr, *d_seeds = nullptr;
	WC_HIP(hipMalloc(&d_host, host.size() * sizeof(uint32_t)));
	hipError_t e = hipMalloc(&d_seeds, (size_t)n_chunks * sizeof(uint4));
	if (e == hipSuccess) e = hipMemcpyAsync(d_host, host.data(), host.size() * sizeof(uint32_t), hipMemcpyHostToDevice, dev->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(rng_seed_expand_kernel, dim3((n_super + 63) / 64), dim3(64), 0, dev->stream, d_host, d_host + n_super, d_seeds, n_chunks);
		hipLaunchKernelGGL(rng_fill_kernel, dim3((n_chunks + 255) / 256), dim3(256), 0, dev->stream, d_seeds, table, n_chunks);
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipStreamSynchronize(dev->stream);  // the host vector must outlive the copy
	(void)hipFree(d_host);
	if (d_seeds) (void)hipFree(d_seeds);
	if (e != hipSuccess) return fail(WC_ERR_DEVICE, std::string("rng_fill: ") + hipGetErrorString(e));
	return WC_OK;
}

int Device::ensure_rng(uint64_t first, uint64_t last) {
	std::lock_guard<std::recursive_mutex> lk(mu);
	if (last <= first) return WC_OK;
	if (rng_count > 0 && first >= rng_base && last <= rng_base + rng_count) return WC_OK;
	uint64_t nb = (first / kRngChunk) * kRngChunk;
	// keep what a typical reference process would also have consumed before `first`
	if (rng_count > 0 && nb >= rng_base && nb - rng_base < (1ull << 26)) nb = rng_base;
	uint64_t ne = ((last + kRngChunk - 1) / kRngChunk) * kRngChunk;
	uint64_t cnt = ne - nb;
	cnt += cnt / 8;  // slack so slightly longer batches do not regenerate
	cnt = ((cnt + kRngChunk - 1) / kRngChunk) * kRngChunk;
	if (cnt > (1ull << 33)) return fail(WC_ERR_UNSUPPORTED, "RNG table request too large");
	quiesce();  // earlier kernels -- on whichever stream -- may still read the old table
	rng_count = 0;
	int rc = rng_table.reserve(cnt * sizeof(uint32_t));
	if (rc) return rc;
	rc = launch_rng_fill(this, rng_table.as<uint32_t>(), nb, cnt);
	if (rc) return rc;
	rng_base = nb;
	rng_count = cnt;
	return WC_OK;
}

int Device::time_begin(const char *name, hipStream_t s) {
	if (!timing) return WC_OK;
	if (!s) s = active();
	const std::string key = time_tag >= 0 ? std::string(name) + "#" + std::to_string(time_tag) : std::string(name);
	auto it = events.find(key);
	if (it == events.end()) {
		hipEvent_t a, b;
		WC_HIP(hipEventCreate(&a));
		WC_HIP(hipEventCreate(&b));
		it = events.emplace(key, std::make_pair(a, b)).first;
	}
	WC_HIP(hipEventRecord(it->second.first, s));
	return WC_OK;
}
int Device::time_end(const char *name, hipStream_t s) {
	if (!timing) return WC_OK;
	if (!s) s = active();
	const std::string key = time_tag >= 0 ? std::string(name) + "#" + std::to_string(time_tag) : std::string(name);
	auto it = events.find(key);
	if (it == events.end()) return WC_OK;
	WC_HIP(hipEventRecord(it->second.second, s));
	return WC_OK;
}

// ------------------------------------------------------------------------------------------------
static std::mutex g_mu;
static std::map<int, std::unique_ptr<Device>> g_devices;
static thread_local int g_device_id = 0;
static thread_local void *g_user_stream = nullptr;  // wc_set_stream: per host thread, never stored in the shared Device
static std::atomic<uint64_t> g_rng_position{0};

uint64_t global_rng_position() { return g_rng_position.load(); }
void set_global_rng_position(uint64_t position) { g_rng_position.store(position); }

hipStream_t Device::active() const { return g_user_stream ? (hipStream_t)g_user_stream : stream; }
void Device::quiesce() const {
	int cur = -1;
	if (hipGetDevice(&cur) == hipSuccess && cur != id) (void)hipSetDevice(id);
	(void)hipDeviceSynchronize();
	if (cur >= 0 && cur != id) (void)hipSetDevice(cur);
}

void Device::release_batch_staging() {
	std::lock_guard<std::recursive_mutex> lk(mu);
	quiesce();  // (a copy out of the staging may still be in flight on a caller's stream)
	for (Staging &st : batch) { st.h.release(); st.d.release(); }
}
void Device::handle_born() {
	std::lock_guard<std::recursive_mutex> lk(mu);
	++live_handles;
}
void Device::handle_gone() {
	std::lock_guard<std::recursive_mutex> lk(mu);
	if (--live_handles <= 0) {
		live_handles = 0;
		// Round 6 (advice of round 5): the batch calls' staging is given back with the last handle only where it is large.  A
		// caller that creates and destroys its stage objects per call -- Python objects going out of scope between batches do
		// exactly that -- would otherwise re-allocate and re-pin it on every wc_*_compute_batch (hipHostMalloc of a gigabyte
		// takes hundreds of milliseconds).  Up to 256 MB stay until wc_release_scratch() or the end of the process.
		size_t held = 0;
		for (const Staging &st : batch) held += st.h.cap + st.d.cap;
		if (held > (size_t)256 << 20) release_batch_staging();
	}
}

OnDeviceOf::OnDeviceOf(const Device *d) : prev(g_device_id) { if (d) g_device_id = d->id; }
OnDeviceOf::~OnDeviceOf() { g_device_id = prev; }

Device *current_device() {
	std::lock_guard<std::mutex> lk(g_mu);
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n <= 0) {
		set_error(std::string("no usable HIP device (hipGetDeviceCount: ") + hipGetErrorString(e) +
				  "); this library has no CPU fallback");
		return nullptr;
	}
	if (g_device_id < 0 || g_device_id >= n) {
		set_error("wc_set_device: device index out of range");
		return nullptr;
	}
	if (hipSetDevice(g_device_id) != hipSuccess) {
		set_error("hipSetDevice failed");
		return nullptr;
	}
	auto it = g_devices.find(g_device_id);
	if (it == g_devices.end()) {
		std::unique_ptr<Device> d(new Device);
		d->id = g_device_id;
		if (hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess) {
			set_error("hipStreamCreate failed");
			return nullptr;
		}
		std::vector<double2> tw(kTwTotal, make_double2(0.0, 0.0));
		for (int k = 0; k < kTwiddleN; ++k) {
			double a = 2.0 * 3.14159265358979323846 * k / kTwiddleN;
			tw[k] = make_double2(std::cos(a), std::sin(a));
		}
		// exact values on the axes / diagonals keep symmetric spectra symmetric
		tw[0] = make_double2(1.0, 0.0);
		tw[kTwiddleN / 4] = make_double2(0.0, 1.0);
		tw[kTwiddleN / 2] = make_double2(-1.0, 0.0);
		tw[3 * kTwiddleN / 4] = make_double2(0.0, -1.0);
		// tables of the wavefront transforms behind the main one (wc_wavefft.hpp): the second stage's twiddles row by row,
		// and the arguments of the lean log / exp
		for (int k = 0; k < 16; ++k)
			for (int r = 0; r < 16; ++r) tw[kTwT2 + 16 * r + k] = tw[(16 * r * k) % kTwiddleN];
		for (int r = 1; r <= 3; ++r)
			for (int j = 0; j < 256; ++j) tw[kTwP3 + 256 * (r - 1) + j] = tw[(4 * r * j) % kTwiddleN];
		for (int n = 0; n <= 1024; ++n) tw[kTwU + n] = tw[2 * n];
		for (int j = 0; j < 1024; ++j) tw[kTwUo + j] = tw[2 * j + 1];
		// the 512-point transform at eight points per lane (wf8_*)
		for (int n1 = 0; n1 < 8; ++n1)
			for (int ka = 0; ka < 8; ++ka) tw[kTw8A + 8 * n1 + ka] = tw[(64 * n1 * ka) % kTwiddleN];
		for (int kb = 0; kb < 8; ++kb)
			for (int t = 0; t < 64; ++t) tw[kTw8B + 64 * kb + t] = tw[(8 * (t & 7) * ((t >> 3) + 8 * kb)) % kTwiddleN];
		for (int n = 0; n <= 512; ++n) tw[kTw8U + n] = tw[4 * n];
		{
			double *is = reinterpret_cast<double *>(&tw[kTwIS]);  // wf_even2048
			is[0] = 0.0;
			for (int k = 1; k < 512; ++k) is[k] = (double)(1.0L / (4.0L * sinl(2.0L * 3.14159265358979323846264338327950288L * k / 2048.0L)));
		}
		for (int i = 0; i < 128; ++i) {
			const double c = 0.5 + (i + 0.5) / 256.0, invc = 1.0 / c;
			tw[kTwLog + i] = make_double2(invc, (double)-logl((long double)invc));
		}
		{
			double *ex = reinterpret_cast<double *>(&tw[kTwExp]);
			for (int j = 0; j < 64; ++j) ex[j] = (double)exp2l((long double)j / 64.0L);
			double *ik = reinterpret_cast<double *>(&tw[kTwInvK]);
			for (int k = 1; k < 1040; ++k) ik[k] = 1.0 / k;
		}
		if (hipMalloc(&d->twiddle, sizeof(double2) * kTwTotal) != hipSuccess ||
			hipMemcpy(d->twiddle, tw.data(), sizeof(double2) * kTwTotal, hipMemcpyHostToDevice) != hipSuccess) {
			set_error("twiddle table upload failed");
			return nullptr;
		}
		it = g_devices.emplace(g_device_id, std::move(d)).first;
	}
	return it->second.get();
}

}  // namespace wc

using namespace wc;

extern "C" {

const char *wc_last_error(void) { return g_error.c_str(); }
const char *wc_version(void) { return "world_class_amd 0.2 (gfx950)"; }
// SHA-256 of the sources and flags this library was built from (world_class_amd/build.py checks it against the tree)
#ifndef WC_SOURCE_HASH
#define WC_SOURCE_HASH "unstamped"
#endif
const char *wc_build_hash(void) {
	static const char stamp[] = "WC_SOURCE_HASH=" WC_SOURCE_HASH;
	return stamp + 15;
}
int wc_device_count(void) {
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess) return fail(WC_ERR_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
	return n;
}
int wc_set_device(int device) {
	if (device < 0) return fail(WC_ERR_INVALID, "negative device index");
	g_device_id = device;
	return WC_OK;
}
int wc_get_device(void) { return g_device_id; }
int wc_set_stream(void *hip_stream) {
	g_user_stream = hip_stream;  // thread-local: other threads and the library's own stream are untouched
	return WC_OK;
}
int wc_synchronize(void) {
	Device *d = current_device();
	if (!d) return WC_ERR_DEVICE;
	WC_HIP(hipStreamSynchronize(d->active()));
	return WC_OK;
}
// frees the staging the host-pointer batch calls keep on the calling thread's device (page-locked host memory and its device
// twin; they come back on the next batch call)
int wc_release_scratch(void) {
	Device *d = current_device();
	if (!d) return WC_ERR_DEVICE;
	d->release_batch_staging();
	return WC_OK;
}
uint64_t wc_rng_get_position(void) { return g_rng_position.load(); }
void wc_rng_set_position(uint64_t position) { g_rng_position.store(position); }

// reference src/world_matlabfunctions.cpp:243-264 as a host function on the same process-wide stream the stages
// use: draws at the current position and advances it by one (declared in include/world_matlabfunctions.hpp)
double randn(void) {
	static std::mutex mu;
	static uint64_t cached_pos = ~0ull;
	static XorShift cached;
	std::lock_guard<std::mutex> lk(mu);
	const uint64_t now = g_rng_position.load();
	if (cached_pos != now) {  // someone moved the stream (a stage ran, or wc_rng_set_position): jump there
		uint32_t st[4];
		rng_state_at(now, st);
		cached = XorShift{st[0], st[1], st[2], st[3]};
	}
	const uint32_t raw = cached.draw();
	cached_pos = now + 1;
	g_rng_position.store(cached_pos);
	return raw / 268435456.0 - 6.0;
}

// reference src/harvest.cpp:173-181
int wc_get_samples(int fs, int x_length, double frame_period) {
	return static_cast<int>(1000.0 * x_length / fs / frame_period) + 1;
}
// reference src/cheaptrick.cpp:97-105
int wc_cheaptrick_fft_size(int fs, double f0_floor) {
	return static_cast<int>(std::pow(2.0, 1.0 + static_cast<int>(std::log(3.0 * fs / f0_floor + 1) / 0.69314718055994529)));
}
double wc_cheaptrick_f0_floor(int fs, int fft_size) { return 3 * fs / (fft_size - 3.0); }
// reference test/test.cpp:362-363
int wc_synthesis_out_length(int f0_length, double frame_period, int fs) {
	return static_cast<int>((f0_length - 1) * frame_period / 1000.0 * fs) + 1;
}

void *wc_device_malloc(uint64_t bytes) {
	if (!current_device()) return nullptr;
	void *p = nullptr;
	hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
	if (e != hipSuccess) {
		set_error(std::string("hipMalloc: ") + hipGetErrorString(e));
		return nullptr;
	}
	return p;
}
void wc_device_free(void *p) {
	if (p) (void)hipFree(p);
}
int wc_memcpy_h2d(void *dst, const void *src, uint64_t bytes) {
	Device *d = current_device();
	if (!d) return WC_ERR_DEVICE;
	WC_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, d->active()));
	WC_HIP(hipStreamSynchronize(d->active()));
	return WC_OK;
}
int wc_memcpy_d2h(void *dst, const void *src, uint64_t bytes) {
	Device *d = current_device();
	if (!d) return WC_ERR_DEVICE;
	WC_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, d->active()));
	WC_HIP(hipStreamSynchronize(d->active()));
	return WC_OK;
}
int wc_set_kernel_timing(int enable) {
	Device *d = current_device();
	if (!d) return WC_ERR_DEVICE;
	if (enable && !d->timing) {  // a fresh measurement window: forget the events of earlier windows
		for (auto &kv : d->events) { (void)hipEventDestroy(kv.second.first); (void)hipEventDestroy(kv.second.second); }
		d->events.clear();
	}
	d->timing = enable != 0;
	return WC_OK;
}
float wc_last_kernel_ms(const char *kernel_name) {
	// sum over the plain event and the per-group events "name#k" of the fused pipeline
	Device *d = current_device();
	if (!d) return -1.f;
	const std::string base(kernel_name);
	float total = 0.f;
	bool any = false;
	for (auto &kv : d->events) {
		if (kv.first != base && kv.first.compare(0, base.size() + 1, base + "#") != 0) continue;
		if (hipEventQuery(kv.second.second) == hipErrorNotReady && hipEventSynchronize(kv.second.second) != hipSuccess) continue;
		float ms = -1.f;
		if (hipEventElapsedTime(&ms, kv.second.first, kv.second.second) != hipSuccess) continue;
		total += ms;
		any = true;
	}
	return any ? total : -1.f;
}

}  // extern "C"
