// Utterance sharding and the final gather for C / C++ hosts: include/world_class_shard.h.  Host code only.
#include <dlfcn.h>

#include <algorithm>
#include <numeric>
#include <vector>

#include "wc_internal.hpp"
#include "../../include/world_class_shard.h"

using namespace wc;

namespace {
// the four RCCL entry points the gather needs, bound at first use
struct Rccl {
	void *handle = nullptr;
	int (*group_start)() = nullptr;
	int (*group_end)() = nullptr;
	int (*broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
	int (*send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
	int (*recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
	const char *(*error_string)(int) = nullptr;
	bool ok = false;
};
Rccl &rccl() {
	static Rccl r;
	static std::once_flag once;
	std::call_once(once, [] {
		// a copy that is already in the process first (a host that created the communicator has one; PyTorch ships its own)
		const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
		for (const char *n : names)
			if ((r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
		if (!r.handle)
			for (const char *n : names)
				if ((r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
		if (!r.handle) return;
		r.group_start = reinterpret_cast<int (*)()>(dlsym(r.handle, "ncclGroupStart"));
		r.group_end = reinterpret_cast<int (*)()>(dlsym(r.handle, "ncclGroupEnd"));
		r.broadcast = reinterpret_cast<int (*)(const void *, void *, size_t, int, int, void *, hipStream_t)>(dlsym(r.handle, "ncclBroadcast"));
		r.send = reinterpret_cast<int (*)(const void *, size_t, int, int, void *, hipStream_t)>(dlsym(r.handle, "ncclSend"));
		r.recv = reinterpret_cast<int (*)(void *, size_t, int, int, void *, hipStream_t)>(dlsym(r.handle, "ncclRecv"));
		r.error_string = reinterpret_cast<const char *(*)(int)>(dlsym(r.handle, "ncclGetErrorString"));
		r.ok = r.group_start && r.group_end && r.broadcast;
	});
	return r;
}
constexpr int kNcclDouble = 8;  // ncclFloat64, rccl.h
}  // namespace

extern "C" {

int wc_shard_partition(const int *lengths, int n, int world, int *rank_of) {
	if (n < 0 || world <= 0 || (n > 0 && (!lengths || !rank_of))) return fail(WC_ERR_INVALID, "shard partition: bad argument");
	std::vector<int> order(n);
	std::iota(order.begin(), order.end(), 0);
	std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return lengths[a] > lengths[b]; });  // ties keep index order
	std::vector<long long> load(world, 0);
	for (int i : order) {
		const int r = (int)(std::min_element(load.begin(), load.end()) - load.begin());  // lightest rank, lowest index on ties
		rank_of[i] = r;
		load[r] += lengths[i];
	}
	return WC_OK;
}

int wc_gather_device(void *nccl_comm, int world, int rank, const double *d_local, const long long *counts, double *d_all) {
	if (!nccl_comm || world <= 0 || rank < 0 || rank >= world || !counts || !d_all) return fail(WC_ERR_INVALID, "gather: bad argument");
	for (int r = 0; r < world; ++r)
		if (counts[r] < 0) return fail(WC_ERR_INVALID, "gather: negative count");
	if (counts[rank] > 0 && !d_local) return fail(WC_ERR_INVALID, "gather: null local buffer");
	Device *dev = current_device();
	if (!dev) return WC_ERR_DEVICE;
	DeviceLock lock(dev);
	Rccl &R = rccl();
	if (!R.ok) return fail(WC_ERR_UNSUPPORTED, "gather: librccl could not be loaded");
	hipStream_t s = dev->active();
	auto check = [&](int rc, const char *what) {
		if (rc == 0) return WC_OK;
		return fail(WC_ERR_DEVICE, std::string("gather: ") + what + ": " + (R.error_string ? R.error_string(rc) : "RCCL error"));
	};
	int rc;
	if ((rc = check(R.group_start(), "ncclGroupStart"))) return rc;
	long long off = 0;
	for (int r = 0; r < world; ++r) {
		// one broadcast per contributing rank, fused in a group: each root's block travels its own xGMI links concurrently
		if (counts[r] > 0) {
			const int e = R.broadcast(r == rank ? d_local : d_all + off, d_all + off, (size_t)counts[r], kNcclDouble, r, nccl_comm, s);
			if (e != 0) {
				(void)R.group_end();
				return check(e, "ncclBroadcast");
			}
		}
		off += counts[r];
	}
	return check(R.group_end(), "ncclGroupEnd");
}

int wc_gather_to_root_device(void *nccl_comm, int world, int rank, int root, const double *d_local, const long long *counts, double *d_all) {
	if (!nccl_comm || world <= 0 || rank < 0 || rank >= world || root < 0 || root >= world || !counts) return fail(WC_ERR_INVALID, "gather to root: bad argument");
	for (int r = 0; r < world; ++r)
		if (counts[r] < 0) return fail(WC_ERR_INVALID, "gather to root: negative count");
	if (counts[rank] > 0 && !d_local) return fail(WC_ERR_INVALID, "gather to root: null local buffer");
	if (rank == root && !d_all) return fail(WC_ERR_INVALID, "gather to root: the root needs the destination");
	Device *dev = current_device();
	if (!dev) return WC_ERR_DEVICE;
	DeviceLock lock(dev);
	Rccl &R = rccl();
	if (!R.ok || !R.send || !R.recv) return fail(WC_ERR_UNSUPPORTED, "gather to root: librccl (ncclSend / ncclRecv) could not be loaded");
	hipStream_t s = dev->active();
	auto check = [&](int rc, const char *what) {
		if (rc == 0) return WC_OK;
		return fail(WC_ERR_DEVICE, std::string("gather to root: ") + what + ": " + (R.error_string ? R.error_string(rc) : "RCCL error"));
	};
	int rc;
	if (rank != root) {  // one send over this rank's own xGMI link to the root
		if (counts[rank] == 0) return WC_OK;
		return check(R.send(d_local, (size_t)counts[rank], kNcclDouble, root, nccl_comm, s), "ncclSend");
	}
	// the root: its own block by a device copy, the others by grouped receives -- every peer's link carries its block at once
	long long off = 0;
	if ((rc = check(R.group_start(), "ncclGroupStart"))) return rc;
	for (int r = 0; r < world; ++r) {
		if (r != root && counts[r] > 0) {
			const int e = R.recv(d_all + off, (size_t)counts[r], kNcclDouble, r, nccl_comm, s);
			if (e != 0) {
				(void)R.group_end();
				return check(e, "ncclRecv");
			}
		}
		off += counts[r];
	}
	if ((rc = check(R.group_end(), "ncclGroupEnd"))) return rc;
	off = 0;
	for (int r = 0; r < root; ++r) off += counts[r];
	if (counts[root] > 0) WC_HIP(hipMemcpyAsync(d_all + off, d_local, sizeof(double) * (size_t)counts[root], hipMemcpyDeviceToDevice, s));
	return WC_OK;
}

}  // extern "C"
