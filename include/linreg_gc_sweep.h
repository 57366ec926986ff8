/*
 * linreg_gc_sweep.h -- the per-lambda sweep (BASELINE config 5) of liblinreg_gc.so.
 *
 * Not part of the drop-in surface of linreg_gc.h: the reference runs one execYaoProtocol per regularisation value
 * (src/cmd/linreg.c:177 inside the wrapper's / the experiments' lambda loops).  lambda is a public constant added to the
 * diagonal AFTER the data providers' shares are summed (src/linear.oc:52-57), so `count` circuits that differ only in
 * lambda share their input labels, the garbled share-summation launches and the division by the public normalizer -- the
 * PREFIX -- and run as one merged program; the blocks of a sweep sharded over several GPUs share that prefix too.
 */
#ifndef LINREG_GC_SWEEP_H
#define LINREG_GC_SWEEP_H
#include "linreg_gc.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- both roles on one GPU (python/sweep.py, bench.py --gpus N: one rank per GPU, RCCL broadcast of the prefix) */
/* Per-lambda sweep (BASELINE config 5): `count` circuits that differ only in the public
 * regularisation constant added to the diagonal (src/linear.oc:52-57), garbled and evaluated as one
 * program.  lambda enters AFTER the shares are summed, so the input labels, the garbled
 * share-summation launches and the division by the public normalizer (off the diagonal and in b:
 * linear.oc:57-65) -- the shared prefix -- exist once for the whole sweep (a data provider
 * runs one label OT whatever the number of lambdas); the launches of all circuits are merged, so the
 * latency-bound stages (dividers, reveals) of different circuits fill the GPU together.
 * sys->lambda is ignored; sys->normalize must be 1, trace and reveal_inputs 0.  All circuits read
 * the shares given to lgc_solver_set_shares; lgc_solver_get_beta returns count x d words
 * (circuit-major). */
int lgc_solver_create_sweep(lgc_solver **out, int device, const lgc_system *sys, const uint8_t seed[16],
                            size_t count, const double *lambdas);
size_t lgc_solver_num_circuits(const lgc_solver *s);
/* One rank's block of a sweep sharded over several GPUs: circuits
 * [first, first + count) of the whole sweep.  All ranks use the SAME seed: they share the prefix, hence
 * the garbler's offset R, and `first` keeps the gate ids of different ranks' circuits disjoint.
 *   rank 0:     lgc_solver_set_shares; lgc_solver_prefix_garble        (input labels; the prefix garbled AND evaluated)
 *               lgc_solver_prefix_export(dev_buf)                      -> broadcast (RCCL over xGMI)
 *   every rank: lgc_solver_prefix_import(dev_buf); lgc_solver_run      (garbles + evaluates its own circuits on the
 *               words the prefix left; the prefix launches are not run again -- rounds 2-5 shipped their tables too)
 * dev_buf: device memory of lgc_solver_prefix_bytes() bytes on the solver's GPU, owned by the caller
 * (e.g. a torch tensor handed to torch.distributed.broadcast).  Layout: garbler words of the shared
 * region | evaluator words of the shared region, both as the prefix leaves them. */
int lgc_solver_create_sweep_at(lgc_solver **out, int device, const lgc_system *sys, const uint8_t seed[16],
                               size_t count, const double *lambdas, size_t first);
size_t lgc_solver_prefix_bytes(const lgc_solver *s);
int lgc_solver_prefix_garble(lgc_solver *s);
int lgc_solver_prefix_export(lgc_solver *s, void *dev_buf);
int lgc_solver_prefix_import(lgc_solver *s, const void *dev_buf);


/* ---- the two roles apart (bin/linreg --lambdas [--devices=g0,g1,...]) */
/* The per-lambda sweep with the roles apart (see lgc_solver_create_sweep): ONE set of input labels and
 * one garbled share summation for all `count` circuits, so every data provider runs its label OT
 * (lgc_ot_labels_*, src/input.c:37-50) once whatever the number of lambdas; lgc_party_finish then
 * returns count x d words of beta (circuit-major).  Both sides pass the same count and lambdas. */
int lgc_party_create_sweep(lgc_party **out, int device, const lgc_system *sys, int role, const uint8_t seed[16],
                           size_t max_launch_table_bytes, size_t count, const double *lambdas);
size_t lgc_party_num_circuits(const lgc_party *p);
/* The sweep on several GPUs of one CSP / Evaluator process (bin/linreg --lambdas --devices=...; SURVEY.md 8(e),
 * src/cmd/linreg.c:145-199 runs one execYaoProtocol per circuit): each device gets ONE party object holding the
 * contiguous block [first, first + count) of the sweep's circuits.  All garbler blocks share the seed -- one set
 * of input labels, one label OT per data provider -- and `first` keeps the gate ids of different blocks disjoint.
 * The prefix launches [0, lgc_party_prefix_launches) (share summation, normalizer; lambda enters after them) are garbled /
 * evaluated by the first block only; lgc_party_share_prefix copies the words they produce to another block of the
 * same role (another GPU: over xGMI), which then runs the launches from lgc_party_prefix_launches on. */
int lgc_party_create_sweep_at(lgc_party **out, int device, const lgc_system *sys, int role, const uint8_t seed[16],
                              size_t max_launch_table_bytes, size_t count, const double *lambdas, size_t first);
size_t lgc_party_prefix_launches(const lgc_party *p);
uint64_t lgc_party_prefix_and_gates(const lgc_party *p);
int lgc_party_share_prefix(lgc_party *dst, const lgc_party *src);

/* Preflight of a device list (bin/linreg --devices, bench.py --gpus N) before anything is allocated: every index exists, and
 * every pair of DISTINCT devices can reach each other (hipDeviceCanAccessPeer, both ways) -- the shared prefix travels by peer
 * copy (lgc_party_share_prefix) and an Evaluator on another GPU maps the CSP's table ring over hipIpc.  An index may repeat
 * (several blocks on one GPU: the one-GPU rehearsal).  LGC_EINVAL with a message that names the index or the pair. */
int lgc_devices_preflight(const int *devices, size_t n);

#ifdef __cplusplus
}
#endif
#endif
