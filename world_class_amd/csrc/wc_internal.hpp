// Host-side internals shared by the C-ABI translation units: per-device state (stream, twiddle
// table, RNG draw table), grow-only device buffers, error reporting, kernel timing.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/world_class_c.h"

namespace wc {

void set_error(const std::string &msg);
int fail(int code, const std::string &msg);

#define WC_HIP(expr)                                                                              \
	do {                                                                                          \
		hipError_t _e = (expr);                                                                   \
		if (_e != hipSuccess)                                                                     \
			return ::wc::fail(WC_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));  \
	} while (0)

// Grow-only device buffer.
struct DevBuf {
	void *p = nullptr;
	size_t cap = 0;
	int reserve(size_t bytes);  // 0 on success
	void release();
	template <class T> T *as() const { return static_cast<T *>(p); }
};

// Pinned host staging buffer (grow-only).
struct HostBuf {
	void *p = nullptr;
	size_t cap = 0;
	hipEvent_t ev = nullptr;  // recorded after the async copies that read this buffer
	int reserve(size_t bytes);  // also waits until earlier async copies out of the buffer are done
	int mark(hipStream_t s);    // call after enqueueing the copies that read the buffer
	void release();
	template <class T> T *as() const { return static_cast<T *>(p); }
};

// host arrays <-> one packed device array: page-locked staging and its device twin (wc_batch.hip)
struct Staging {
	HostBuf h;
	DevBuf d;
};

// Per-device shared state.
struct Device {
	int id = 0;
	hipStream_t stream = nullptr;  // library-owned non-blocking stream: created once, never replaced
	// the stream this thread's calls enqueue on: the calling thread's wc_set_stream() stream, else `stream`
	hipStream_t active() const;
	// waits for everything on the device (handles are destroyed / the RNG table is replaced behind it: work that uses
	// their buffers may sit on the library stream, on a caller's stream or on the pipeline's side streams)
	void quiesce() const;
	// One public call at a time per device: the calls of several host threads that drive handles on the same device
	// are serialised (they share the RNG draw table, the timing events and the noise-stream position).
	std::recursive_mutex mu;
	double2 *twiddle = nullptr;  // kTwiddleN entries, e^{+2 pi i k / kTwiddleN}
	// RNG draw table: raw 12-step sums (uint32) of the reference's randn() for stream positions
	// [rng_base, rng_base + rng_count)
	DevBuf rng_table;
	uint64_t rng_base = 0, rng_count = 0;
	int ensure_rng(uint64_t first, uint64_t last_exclusive);  // makes [first, last) available
	// timing
	bool timing = false;
	int time_tag = -1;  // >= 0: events are keyed "name#tag" (the pipeline runs each kernel once per utterance group)
	std::map<std::string, std::pair<hipEvent_t, hipEvent_t>> events;
	int time_begin(const char *name, hipStream_t s = nullptr);  // nullptr = active()
	int time_end(const char *name, hipStream_t s = nullptr);
	// Staging of the host-pointer batch calls (wc_*_compute_batch): samples, time axis, contour, two row matrices, waveform.
	// Owned by the device, used under its call lock by whichever host thread calls, grown on demand and kept between calls
	// (a 64 x 10 s batch at 48 kHz holds ~1 GB of device and ~1 GB of page-locked memory per row matrix).  Released when the
	// last stage handle on the device is destroyed and by wc_release_scratch() -- never tied to a host thread's lifetime.
	Staging batch[6];
	int live_handles = 0;  // stage handles alive on this device (under mu)
	void handle_born();
	void handle_gone();  // the last one takes the batch staging with it
	void release_batch_staging();
};

Device *current_device();  // creates the state on first use; nullptr + error on failure
// RAII: a handle that creates a helper handle of its own on first use (a twin, a pipeline group) does so on ITS device, whatever the
// calling thread's wc_set_device says (thread-local; the compute entry points run a handle on its own device anyway)
struct OnDeviceOf {
	int prev;
	explicit OnDeviceOf(const Device *d);
	~OnDeviceOf();
};
// process-wide noise-stream position of the host-pointer calls (atomic: handles on different devices share it)
uint64_t global_rng_position();
void set_global_rng_position(uint64_t position);
// RAII: serialises the public calls on one device
struct DeviceLock {
	std::unique_lock<std::recursive_mutex> lk;
	explicit DeviceLock(Device *d) : lk(d->mu) {}
};

// utterance descriptors uploaded per batch call
struct UttDesc {
	long long x_off;   // first sample in the packed sample array
	long long f_off;   // first frame (row) in the packed frame arrays
	long long y_off;   // first output sample (synthesis)
	int x_len, f_len, y_len, pad;
	unsigned long long rng_pos;  // stream position at which this utterance starts the stage
};

// host-side RNG jump-ahead (GF(2) linear algebra on the xorshift128 state)
void rng_state_at(uint64_t position, uint32_t state[4]);

// device entry points implemented in the .hip files
int launch_rng_fill(Device *dev, uint32_t *table, uint64_t first, uint64_t count);

}  // namespace wc
