/*
 * locus_hip_rccl.h -- the two exchange steps of the multi-GPU path on RCCL over xGMI, for a C/C++ host that
 * runs one process (or thread) per GPU without torch.distributed.  liblocus_hip_rccl.so links librccl and liblocus_hip.
 *
 *   1. independent scan pairs sharded over ranks (BASELINE configs 4): the data path has NO collective; the only exchange is
 *      the gather of the per-pair lh_gicp_result records (96 B each)           -> lh_rccl_allgather_results
 *   2. one huge pair sharded by SOURCE points (config 5): one SUM all-reduce of the 74 moment sums per outer iteration
 *      (cost_mode 1) or of the 14 cost sums per evaluation (cost_mode 0)        -> lh_rccl_install_sum_hook (lh_set_allreduce)
 *
 * Bootstrap is the caller's: rank 0 calls lh_rccl_get_unique_id and ships the 128 bytes to the other ranks with whatever the
 * launcher offers (MPI_Bcast, a torch store, a file); every rank then calls lh_rccl_create (= ncclCommInitRank).
 * A single process that drives several GPUs needs none of this: see lh_gicp_align_batch_multi in locus_hip.h.
 */
#ifndef LOCUS_HIP_RCCL_H_
#define LOCUS_HIP_RCCL_H_

#include "locus_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define LH_RCCL_ID_BYTES 128 /* = NCCL_UNIQUE_ID_BYTES */
typedef struct lh_rccl lh_rccl;

lh_status lh_rccl_get_unique_id(char id[LH_RCCL_ID_BYTES]);
/* ncclCommInitRank on `device_id`; collective over the `world` ranks that pass the same id */
lh_status lh_rccl_create(int device_id, const char id[LH_RCCL_ID_BYTES], int rank, int world, lh_rccl** out);
void lh_rccl_destroy(lh_rccl* r);
int lh_rccl_rank(const lh_rccl* r);
int lh_rccl_world(const lh_rccl* r);

/* the lh_allreduce_fn of lh_set_allreduce (user = the lh_rccl*): in-place SUM of n doubles over the ranks */
int lh_rccl_sum_hook(double* sums, int n, void* user);
/* the lh_device_allreduce_fn of lh_set_device_allreduce (user = the lh_rccl*): ncclAllReduce(SUM) of n doubles IN HBM, enqueued on `stream`
   (the stream the library's iteration is queued on); returns without synchronising */
int lh_rccl_device_sum_hook(double* dev_sums, int n, void* stream, void* user);
/* lh_set_allreduce(ctx, lh_rccl_sum_hook, r) + lh_set_device_allreduce(ctx, lh_rccl_device_sum_hook, r): a source-sharded pair then runs the
   device-driven loop with the exchange on the device (one all-reduce per outer iteration, no host copy in the loop); r == NULL
   removes both hooks */
lh_status lh_rccl_install_sum_hook(lh_ctx* ctx, lh_rccl* r);

/* every rank contributes n_local results (the counts may differ); `all` (capacity `cap` records) receives the concatenation
   in rank order on every rank, counts[world] (nullable) the per-rank counts.  LH_EINVAL -- on EVERY rank, before the record
   exchange -- if the total exceeds the smallest `cap` any rank passed (the capacities travel with the counts, so the decision
   is collective and no rank is left waiting in the second all-gather). */
lh_status lh_rccl_allgather_results(lh_rccl* r, const lh_gicp_result* local, int n_local, lh_gicp_result* all, int cap, int* counts);
/* max over the ranks (the timed region of a multi-rank run) and a barrier */
lh_status lh_rccl_max_double(lh_rccl* r, double* v);
lh_status lh_rccl_barrier(lh_rccl* r);

#ifdef __cplusplus
}
#endif
#endif /* LOCUS_HIP_RCCL_H_ */
