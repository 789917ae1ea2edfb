// Variant STREAM of the banded-Cholesky ADMM solver (laxMPC / equMPC): ONE LANE PER INSTANCE.
//
// 64 independent instances per wavefront run the reference's iteration in lock step; every
// floating-point accumulation is performed in the reference's own order
// (formulations/+laxMPC/code_laxMPC_ADMM_C.c:308-633, equMPC: code_equMPC_ADMM_C.c), so with
// EXACT = true (no FMA contraction) the results are bit-identical to the generated C built by
// gcc -O3 on x86-64.  Controller constants are wave-uniform: they are fetched with scalar loads
// (s_load) and fed to v_fma_f64 / v_mul_f64 as SGPR operands - no LDS, no lane idles.
//
// State that has to survive between iterations (v, lambda) and between the two sweeps of one
// iteration (the forward-substituted y) does not fit on chip at 64 instances per wave
// (600 doubles per instance), so it is streamed through HBM in a structure-of-arrays scratch
// [element][instance]: every access is a fully coalesced 512-byte wave transaction.
//
// Per iteration and instance (n=12, m=2, N=15): reads 2 x (v, lambda) + y, writes y + (v, lambda)
// = (4*420 + 2*180) * 8 B = 16.3 KB  ->  HBM-bound kernel.
#pragma once
#include "common.hpp"
#include "tv_update_kernel.inc"  // time-varying solvers: row layouts of their scratch (TvLayout, FistaTvLayout) and the update-phase kernels

#include "admm_stream_kernel.inc"  // the kernels (also the source hiprtc specialises for other plant sizes)
