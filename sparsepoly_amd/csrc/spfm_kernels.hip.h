// spfm_kernels.hip.h -- gfx950 device code of the sparse-FM proximal CD core.
//
// Execution model: the coordinate order is partitioned into
// batches of columns that share no row; one batch = one dependent step.  Two engines:
//   persistent row-block pass (pcd_prb_kernel / lin_prb_kernel): one launch per
//     component pass, each workgroup owns a block of rows, the per-step column partial
//     sums are exchanged as tagged 8-byte granules (agent-scope stores / loads);
//   multi-kernel (multi-GPU, pbcd): per step a gather kernel (workgroup per column, f64
//     wave-shuffle + LDS reduction), [RCCL all-reduce], and a fused chain + scatter
//     kernel; kernel boundaries on one stream are the only inter-workgroup
//     synchronisation and the launch sequence is replayed from a hipGraph.
//
// The minibatch solver (psgd, spfm_psgd.hip.h) is throughput-bound instead: rows of a
// batch in parallel, then dense HBM passes over P for the step and the prox.
//
// Storage type T (float|double): X values, A caches, (yhat,y).  Everything that
// is reduced or fed to the prox is float64.
#pragma once
#include "spfm_common.hip.h"
#include "spfm_pcd.hip.h"
#include "spfm_prb.hip.h"
#include "spfm_linear.hip.h"
#include "spfm_pbcd.hip.h"
#include "spfm_pbprb.hip.h"
#include "spfm_pcdw.hip.h"
#include "spfm_predict.hip.h"
#include "spfm_psgd.hip.h"
