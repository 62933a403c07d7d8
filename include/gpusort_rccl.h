/* gpusort_rccl.h -- BASELINE configs[4]: the MSB sort sharded over the GPUs of one node, host side in C++ behind a C ABI
 * (libgpusort_rccl.so = this entry point + RCCL; the kernels stay in libgpusort.so, which has no RCCL dependency).
 *
 * north_star: "The MSB recursive-bucket path partitions top-level buckets across the 8 GPUs of one node with a single
 * RCCL all-to-all over xGMI after the first digit pass"; SURVEY.md 8b lists gs_msb_sort_u32_sharded, 8e the steps.  The
 * reference is single-GPU (msb/tests/main.cu:30-31), so there is no reference interface to cite beyond
 * rdxsrt_unstable_sort's buffer conventions (msb/src/sort/gpu_radix_sort.h:197): caller-owned device arrays, u32 keys,
 * optional u32 values, ascending, unstable.
 *
 * One process per GPU.  Every rank calls gs_msb_sort_u32_sharded with its shard; rank r ends with the r-th slice of the
 * global order in d_keys_out[0 .. *num_out).  Steps (all on `stream`, one host synchronisation to read the gathered
 * bucket sizes):
 *   gs_msb_first_pass_u32 -> ncclAllGather of the 256 bucket sizes -> the same monotone bucket -> rank map on every
 *   rank (balanced totals) -> ONE grouped exchange, ncclGroupStart / world x (ncclSend + ncclRecv) / ncclGroupEnd, no
 *   message above 768 MiB (bigger ones in rounds cut at the same places by sender and receiver: RCCL of ROCm 7.2 was
 *   seen to deliver only half of a >= 2 GiB message to self) -> gs_msb_finish_u32 on what arrived.
 * Returns 0, a hipError_t value, 1000 + ncclResult_t for an RCCL failure, or GS_SHARDED_IMBALANCED when the 256
 * top-byte buckets cannot be cut into `world` ranges that fit `capacity` (a top byte holding most of the keys): the
 * caller then falls back to gs_shard_histogram_u32 / gs_shard_partition_u32 (finer bins) as gpu-sort_amd/sharded.py does. */
#ifndef GPUSORT_RCCL_H_
#define GPUSORT_RCCL_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define GS_SHARDED_IMBALANCED 2001
#define GS_SHARDED_PEER_FAILED 2002   /* another rank reported a local failure before the exchange; no data was moved */
#define GS_SHARDED_TRUNCATED   2003   /* gs_sharded_selftest: the communicator lost or altered part of a large message */

/* workspace for a rank that holds num_items keys and may receive up to `capacity` */
size_t gs_msb_sharded_temp_bytes(uint64_t num_items, uint64_t capacity, int has_values, int world);

/* d_keys_in / d_vals_in: this rank's shard (d_vals_in NULL = keys only); d_grouped_*: scratch of num_items elements;
 * d_recv_* and d_*_out: `capacity` elements each (capacity >= the rank's share; 1.25 x the mean is plenty for
 * uniform keys).  nccl_comm: an ncclComm_t of `world` ranks whose rank `rank` is this process.                        */
int gs_msb_sort_u32_sharded(void *d_temp, size_t temp_bytes, const uint32_t *d_keys_in, const uint32_t *d_vals_in,
                            uint64_t num_items, uint32_t *d_grouped_keys, uint32_t *d_grouped_vals,
                            uint32_t *d_recv_keys, uint32_t *d_recv_vals, uint32_t *d_keys_out, uint32_t *d_vals_out,
                            uint64_t capacity, uint64_t *num_out, void *nccl_comm, int rank, int world, int key_type,
                            void *stream);

/* Error behaviour of gs_msb_sort_u32_sharded.  A rank whose first pass fails LOCALLY still takes part in the size
 * all-gather (with a marker instead of sizes), so every rank returns -- the failing one with its error, the others with
 * GS_SHARDED_PEER_FAILED -- and nobody is left waiting in a collective.  A failure INSIDE the exchange (ncclSend /
 * ncclRecv / ncclGroupEnd) always closes the group it opened, then aborts the communicator (ncclCommAbort: the
 * handle is dead afterwards) so that the peers' pending operations end instead of waiting for this rank for ever.   */

/* The exchange plan of one rank, a pure host function of the gathered sizes (exposed for tests; every rank runs it on
 * the same `counts` and `dest`): send_off[world + 1] = where rank r's slice of MY grouped shard starts (element
 * offsets), recv_off[world + 1] = where source r's piece starts in MY receive buffer, pieces[world][256] = what
 * source r sends me per top byte (gs_msb_finish_u32's piece table), *rounds = how many <= 768 MiB rounds the largest
 * (source, destination) message of the whole exchange needs -- the same number on every rank.                        */
void gs_sharded_exchange_plan(const uint64_t *counts, const uint8_t *dest_of_bucket, int rank, int world,
                              uint64_t *send_off, uint64_t *recv_off, uint64_t *pieces, uint64_t *rounds);

/* Start-up check of a communicator: every rank sends ONE message of `elements` 32-bit words (use 3 << 26 = 768 MiB, the
 * largest message the sort emits) to the next rank of a ring (to itself when world == 1) and compares what arrived
 * word for word.  Returns 0, a hipError_t / 1000 + ncclResult_t value, or GS_SHARDED_TRUNCATED.  RCCL of ROCm 7.2 was
 * seen to deliver only the first half of a self-send of 1 GiB + 4 bytes or more (1 GiB is whole): a machine on which
 * this test fails at 768 MiB must not run the sharded sort.                                                          */
int gs_sharded_selftest(void *nccl_comm, int rank, int world, uint64_t elements, void *stream);

/* the split every rank computes from the gathered sizes (exposed for tests): counts[world][256] -> dest_of_bucket[256],
 * per_rank[world]; bucket b goes to rank floor(world * keys_before_b / n), made monotone.                              */
void gs_sharded_compute_splits(const uint64_t *counts, int world, uint8_t *dest_of_bucket, uint64_t *per_rank);

#ifdef __cplusplus
}
#endif
#endif /* GPUSORT_RCCL_H_ */
