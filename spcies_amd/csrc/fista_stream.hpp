// Variant STREAM of the banded-Cholesky FISTA solver (laxMPC / equMPC): ONE LANE PER INSTANCE, the
// reference's operation order (formulations/+laxMPC/code_laxMPC_FISTA_C.c:296-389 and helpers :471-651,
// equMPC: code_equMPC_FISTA_C.c), no FMA contraction with EXACT = true -> bit-identical results.
//
// One iteration is two sweeps over the N blocks:
//   forward : z(y) block by block (:471-539), residual r = b - G z (:546-574), exit test on |r|
//             (:330-344) and - speculatively, it only writes scratch - the forward substitution of
//             solve_W (:582-612);
//   backward: back substitution (:614-649), lambda = y + d, y = lambda + (t1-1)(lambda - lambda1)/t (:360-385).
// State streamed through the structure-of-arrays scratch [row][instance]: y, lambda, the forward-
// substituted d (N*n rows each) - (4 reads + 3 writes) * N*n * 8 B = 10 KB per iteration at C2.
#pragma once
#include "admm_stream.hpp"

#include "fista_stream_kernel.inc"  // the kernels (also the source hiprtc specialises for other plant sizes)
