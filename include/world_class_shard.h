/* world_class_shard.h -- utterance sharding across the GPUs of one node and the final gather, for C / C++ hosts
 * (the Python mirror world_class_amd/shard.py does the same over torch.distributed).
 *
 * Every utterance is independent in all four stages (the reference's only cross-call state, the randn() position of
 * reference src/world_matlabfunctions.cpp:243-264, is explicit per utterance here), so ranks never exchange data on the
 * data path.  The one collective is the gather of results at the end; it runs on the RCCL communicator the caller owns.
 */
#ifndef WORLD_CLASS_SHARD_H
#define WORLD_CLASS_SHARD_H

#include "world_class_c.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Static longest-processing-time partition of n utterances over `world` ranks: utterances sorted by length (ties by index)
 * are dealt to the currently lightest rank.  Deterministic, identical on every rank; rank_of[i] receives utterance i's rank.
 * Within a rank, utterances keep their original order (the packed batch layout of world_class_c.h). */
int wc_shard_partition(const int *lengths, int n, int world, int *rank_of);

/* All-gather of ragged per-rank results over RCCL (xGMI inside a node): every rank contributes counts[rank] doubles at
 * d_local; afterwards every rank holds the concatenation in rank order at d_all (sum(counts) doubles).  nccl_comm is an
 * ncclComm_t (as void*) created by the caller; counts is a host array identical on all ranks.  Enqueued on the calling
 * thread's stream (wc_set_stream) and ordered after the library's work there; returns without waiting for completion --
 * wc_synchronize() or the caller's own stream synchronisation does that.  librccl is loaded on first use (the copy
 * already in the process, e.g. PyTorch's, is preferred), so hosts that never gather do not need it. */
int wc_gather_device(void *nccl_comm, int world, int rank, const double *d_local, const long long *counts, double *d_all);

/* Gather to ONE rank (what a node that hands its results to one consumer needs): rank r sends its
 * counts[r] doubles to `root` with one ncclSend; the root posts the world - 1 ncclRecv in one group -- xGMI is point to point,
 * every peer has its own link to the root, so the blocks travel concurrently (BASELINE config 4: 0.49 GB per peer, ~3 ms at
 * 153 GB/s per link) -- and copies its own block on the device.  Only the root allocates the total: d_all (sum(counts)
 * doubles, rank order) is used on the root alone and may be NULL elsewhere.  Enqueue-only, like wc_gather_device. */
int wc_gather_to_root_device(void *nccl_comm, int world, int rank, int root, const double *d_local, const long long *counts, double *d_all);

#ifdef __cplusplus
}
#endif
#endif /* WORLD_CLASS_SHARD_H */
