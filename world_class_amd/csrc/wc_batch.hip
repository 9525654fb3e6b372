// Host-pointer BATCH entry points of the four stages (include/world_class_c.h: wc_*_compute_batch; SURVEY.md section 8(b):
// "_compute_batch (extension: n_utt, arrays of pointers / lengths -- the only way to fill an MI355X)").  A C or C++ caller
// that holds its utterances as separate host arrays -- the reference's own calling convention, one compute() per utterance
// (reference include/harvest.hpp:37-39, cheaptrick.hpp:30-33, d4c.hpp:30-34, synthesis.hpp:45-49) -- hands all of them over
// at once: the arrays are packed into page-locked staging, cross PCIe once each way, and the stage's kernels see one batch
// (wc_*_compute_device).  Rows of spectrogram / aperiodicity are the caller's, one pointer per frame, as in the reference.
// Noise stream: rng_pos[u] in / out per utterance like the device calls; NULL = every utterance as in a fresh process.
#include <cstring>
#include <vector>

#include "wc_stages.hpp"
#include "wc_hostcopy.hpp"

using namespace wc;

namespace {

// the device's staging (wc_internal.hpp: Device::batch), named; steady-state calls allocate nothing
struct BatchScratch {
	Staging &x, &t, &f, &a, &b, &y;
};
BatchScratch scratch(Device *dev) {
	return BatchScratch{dev->batch[0], dev->batch[1], dev->batch[2], dev->batch[3], dev->batch[4], dev->batch[5]};
}

// host arrays -> one packed device array (through page-locked staging)
int pack_up(Staging &st, int n, const double *const *src, const int *len, long long *total_out, hipStream_t s) {
	long long total = 0;
	for (int u = 0; u < n; ++u) {
		if (len[u] < 0 || (len[u] > 0 && !src[u])) return fail(WC_ERR_INVALID, "batch: null array or negative length");
		total += len[u];
	}
	*total_out = total;
	if (total == 0) return WC_OK;
	int rc;
	if ((rc = st.h.reserve(sizeof(double) * total))) return rc;
	if ((rc = st.d.reserve(sizeof(double) * total))) return rc;
	std::vector<CopyJob> jobs;
	long long o = 0;
	for (int u = 0; u < n; ++u) {
		if (len[u]) jobs.push_back({st.h.as<double>() + o, src[u], sizeof(double) * (size_t)len[u]});
		o += len[u];
	}
	parallel_copy(jobs);
	WC_HIP(hipMemcpyAsync(st.d.p, st.h.p, sizeof(double) * total, hipMemcpyHostToDevice, s));
	return st.h.mark(s);
}
// rows of all utterances ([utt][frame] pointers to `bins` doubles) -> one packed device array
int pack_rows_up(Staging &st, int n, const double *const *const *rows, const int *f0_length, int bins, hipStream_t s) {
	long long frames = 0;
	for (int u = 0; u < n; ++u) frames += f0_length[u];
	if (frames == 0) return WC_OK;
	int rc;
	if ((rc = st.h.reserve(sizeof(double) * frames * bins))) return rc;
	if ((rc = st.d.reserve(sizeof(double) * frames * bins))) return rc;
	long long o = 0;
	for (int u = 0; u < n; ++u) {
		if (f0_length[u] > 0 && !rows[u]) return fail(WC_ERR_INVALID, "batch: null row table");
		if ((rc = rows_up(s, rows[u], f0_length[u], bins, st.h.as<double>() + o * bins, st.d.as<double>() + o * bins))) return rc;
		o += f0_length[u];
	}
	return st.h.mark(s);
}
// one packed device array -> the caller's arrays
int unpack_down(Staging &st, int n, double *const *dst, const int *len, hipStream_t s) {
	long long total = 0;
	for (int u = 0; u < n; ++u) total += len[u];
	if (total == 0) return WC_OK;
	int rc;
	if ((rc = st.h.reserve(sizeof(double) * total))) return rc;
	WC_HIP(hipMemcpyAsync(st.h.p, st.d.p, sizeof(double) * total, hipMemcpyDeviceToHost, s));
	WC_HIP(hipStreamSynchronize(s));
	std::vector<CopyJob> jobs;
	long long o = 0;
	for (int u = 0; u < n; ++u) {
		if (len[u]) jobs.push_back({dst[u], st.h.as<double>() + o, sizeof(double) * (size_t)len[u]});
		o += len[u];
	}
	parallel_copy(jobs);
	return WC_OK;
}
int unpack_rows_down(Staging &st, int n, double *const *const *rows, const int *f0_length, int bins, hipStream_t s) {
	long long frames = 0;
	for (int u = 0; u < n; ++u) {
		if (f0_length[u] > 0 && !rows[u]) return fail(WC_ERR_INVALID, "batch: null row table");
		frames += f0_length[u];
	}
	if (frames == 0) return WC_OK;
	int rc;
	if ((rc = st.h.reserve(sizeof(double) * frames * bins))) return rc;
	return rows_down_many(s, n, rows, f0_length, bins, st.d.as<double>(), st.h.as<double>());
}

}  // namespace

extern "C" {

int wc_harvest_compute_batch(wc_harvest *h, int n_utt, const double *const *x, const int *x_length,
							 double *const *temporal_positions, double *const *f0) {
	if (!h || n_utt <= 0 || !x || !x_length || !temporal_positions || !f0) return fail(WC_ERR_INVALID, "harvest batch: null argument");
	Device *dev = hv_device(h);  // (the handle's device, whatever the calling thread's wc_set_device says)
	WC_HIP(hipSetDevice(dev->id));
	DeviceLock lock(dev);
	hipStream_t s = dev->active();
	BatchScratch sc = scratch(dev);
	std::vector<int> fl(n_utt);
	long long frames = 0, samples = 0;
	for (int u = 0; u < n_utt; ++u) {
		if (x_length[u] <= 0) return fail(WC_ERR_INVALID, "harvest batch: non-positive length");
		fl[u] = wc_harvest_get_samples(h, x_length[u]);
		frames += fl[u];
	}
	int rc;
	if ((rc = pack_up(sc.x, n_utt, x, x_length, &samples, s))) return rc;
	if ((rc = sc.t.d.reserve(sizeof(double) * frames))) return rc;
	if ((rc = sc.f.d.reserve(sizeof(double) * frames))) return rc;
	if ((rc = wc_harvest_compute_device(h, n_utt, sc.x.d.as<double>(), x_length, sc.t.d.as<double>(), sc.f.d.as<double>()))) return rc;
	if ((rc = unpack_down(sc.t, n_utt, temporal_positions, fl.data(), s))) return rc;
	return unpack_down(sc.f, n_utt, f0, fl.data(), s);
}

int wc_cheaptrick_compute_batch(wc_cheaptrick *c, int n_utt, const double *const *x, const int *x_length,
								const double *const *temporal_positions, const double *const *f0, const int *f0_length,
								double *const *const *spectrogram, uint64_t *rng_pos) {
	if (!c || n_utt <= 0 || !x || !x_length || !temporal_positions || !f0 || !f0_length || !spectrogram)
		return fail(WC_ERR_INVALID, "cheaptrick batch: null argument");
	Device *dev = ct_device(c);  // (the handle's device, whatever the calling thread's wc_set_device says)
	WC_HIP(hipSetDevice(dev->id));
	DeviceLock lock(dev);
	hipStream_t s = dev->active();
	BatchScratch sc = scratch(dev);
	const int bins = wc_cheaptrick_get_fft_size(c) / 2 + 1;
	long long samples = 0, frames = 0, frames2 = 0;
	int rc;
	if ((rc = pack_up(sc.x, n_utt, x, x_length, &samples, s))) return rc;
	if ((rc = pack_up(sc.t, n_utt, temporal_positions, f0_length, &frames, s))) return rc;
	if ((rc = pack_up(sc.f, n_utt, f0, f0_length, &frames2, s))) return rc;
	if (frames == 0) return WC_OK;
	if ((rc = sc.a.d.reserve(sizeof(double) * frames * bins))) return rc;
	if ((rc = wc_cheaptrick_compute_device(c, n_utt, sc.x.d.as<double>(), x_length, sc.t.d.as<double>(), sc.f.d.as<double>(), f0_length,
										   sc.a.d.as<double>(), rng_pos))) return rc;
	return unpack_rows_down(sc.a, n_utt, spectrogram, f0_length, bins, s);
}

int wc_d4c_compute_batch(wc_d4c *d, int n_utt, const double *const *x, const int *x_length, const double *const *temporal_positions,
						 const double *const *f0, const int *f0_length, int fft_size, double *const *const *aperiodicity, uint64_t *rng_pos) {
	if (!d || n_utt <= 0 || !x || !x_length || !temporal_positions || !f0 || !f0_length || !aperiodicity)
		return fail(WC_ERR_INVALID, "d4c batch: null argument");
	if (fft_size < 2 || (fft_size & 1)) return fail(WC_ERR_INVALID, "d4c batch: fft_size must be even and positive");
	Device *dev = d4c_device(d);  // (the handle's device, whatever the calling thread's wc_set_device says)
	WC_HIP(hipSetDevice(dev->id));
	DeviceLock lock(dev);
	hipStream_t s = dev->active();
	BatchScratch sc = scratch(dev);
	const int bins = fft_size / 2 + 1;
	long long samples = 0, frames = 0, frames2 = 0;
	int rc;
	if ((rc = pack_up(sc.x, n_utt, x, x_length, &samples, s))) return rc;
	if ((rc = pack_up(sc.t, n_utt, temporal_positions, f0_length, &frames, s))) return rc;
	if ((rc = pack_up(sc.f, n_utt, f0, f0_length, &frames2, s))) return rc;
	if (frames == 0) return WC_OK;
	if ((rc = sc.a.d.reserve(sizeof(double) * frames * bins))) return rc;
	if ((rc = wc_d4c_compute_device(d, n_utt, sc.x.d.as<double>(), x_length, sc.t.d.as<double>(), sc.f.d.as<double>(), f0_length, fft_size,
									sc.a.d.as<double>(), rng_pos))) return rc;
	return unpack_rows_down(sc.a, n_utt, aperiodicity, f0_length, bins, s);
}

int wc_synthesis_compute_batch(wc_synthesis *sy, int n_utt, const double *const *f0, const int *f0_length, int fft_size,
							   const double *const *const *spectrogram, const double *const *const *aperiodicity, const int *out_length,
							   double *const *out, uint64_t *rng_pos) {
	if (!sy || n_utt <= 0 || !f0 || !f0_length || !spectrogram || !aperiodicity || !out_length || !out)
		return fail(WC_ERR_INVALID, "synthesis batch: null argument");
	// (the rows are packed with the caller's fft_size and read by kernels that stride with the handle's: they must agree)
	if (fft_size != wc_synthesis_get_fft_size(sy)) return fail(WC_ERR_INVALID, "synthesis batch: fft_size differs from the handle's");
	Device *dev = syn_device(sy);  // (the handle's device, whatever the calling thread's wc_set_device says)
	WC_HIP(hipSetDevice(dev->id));
	DeviceLock lock(dev);
	hipStream_t s = dev->active();
	BatchScratch sc = scratch(dev);
	const int bins = fft_size / 2 + 1;
	long long frames = 0, total_out = 0;
	int rc;
	for (int u = 0; u < n_utt; ++u) {
		if (out_length[u] < 0) return fail(WC_ERR_INVALID, "synthesis batch: negative out_length");
		total_out += out_length[u];
	}
	if ((rc = pack_up(sc.f, n_utt, f0, f0_length, &frames, s))) return rc;
	if ((rc = pack_rows_up(sc.a, n_utt, spectrogram, f0_length, bins, s))) return rc;
	if ((rc = pack_rows_up(sc.b, n_utt, aperiodicity, f0_length, bins, s))) return rc;
	if (total_out == 0) return WC_OK;
	if ((rc = sc.y.d.reserve(sizeof(double) * total_out))) return rc;
	if ((rc = wc_synthesis_compute_device(sy, n_utt, sc.f.d.as<double>(), f0_length, sc.a.d.as<double>(), sc.b.d.as<double>(), out_length,
										  sc.y.d.as<double>(), rng_pos))) return rc;
	return unpack_down(sc.y, n_utt, out, out_length, s);
}

}  // extern "C"
