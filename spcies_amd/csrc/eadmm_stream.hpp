// Variant STREAM of the MPCT EADMM solver (diagonal Q, R): ONE LANE PER INSTANCE, the reference's
// operation order (formulations/+MPCT/code_MPCT_EADMM_C.c:85-457), no FMA contraction -> bit-identical.
//
// One iteration = P1 (z1, clamp) + P2 (z2 = W2 q2) in one pass over the stages, then the P3 banded
// solve as a forward and a backward sweep; the backward sweep also forms z3, the residual, the dual
// update and the three-part exit test.  State streamed through the SoA scratch [row][instance]:
// z1, z3 ((N+1)(n+m) rows), lambda ((N+3)(n+m) rows), forward-substituted mu (N n rows); z2 stays in
// registers.
#pragma once
#include "admm_stream.hpp"

#include "eadmm_stream_kernel.inc"  // the kernels (also the source hiprtc specialises for other plant sizes)
