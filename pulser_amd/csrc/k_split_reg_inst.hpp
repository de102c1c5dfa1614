// Part of librydemu: the instantiations of k_split_reg<N, NR, DECAY, ROWS, CPLX, SNAP> the host code launches
// (host_split.hpp: launch_split_reg), one list per register size.  The fully unrolled stage body makes these the most
// expensive kernels of the library to compile (3.4 of the 3.7 minutes of the former single translation unit, device
// side), so with -DRYD_SPLIT_TUS the main unit (rydemu.hip) only DECLARES them (extern template) and three part units
// (rydemu_splitreg.hip, -DRYD_SPLITR_N=12 / 13 / 14) define them, compiled in parallel (__graft_entry__.build, Makefile).
// Without the macro rydemu.hip instantiates them implicitly as before (one command: tools/build_variant.sh, make asan).
#pragma once

#define SPLITR_INSTANCES_OF(N_, X)    \
  X(N_, 5, false, false, false, false) \
  X(N_, 5, true, false, false, false)  \
  X(N_, 5, false, true, false, false)  \
  X(N_, 5, false, false, true, false)  \
  X(N_, 5, true, false, true, false)   \
  X(N_, 5, false, false, false, true)

// 12 atoms, 16 amplitudes per lane on 256 lanes (NR = 4: four waves per sequence instead of two - round 5, batches that
// leave SIMDs idle with NR = 5; launch_split_reg chooses by the grid)
#define SPLITR_INSTANCES_12(X) SPLITR_INSTANCES_OF(12, X) X(12, 4, false, false, false, false) X(12, 4, false, false, false, true)
#define SPLITR_INSTANCES_13(X) SPLITR_INSTANCES_OF(13, X)
#define SPLITR_INSTANCES_14(X) SPLITR_INSTANCES_OF(14, X) X(14, 6, false, false, false, false)

#define SPLITR_EXTERN(N_, NR_, D_, R_, C_, S_) \
  extern template __global__ void k_split_reg<N_, NR_, D_, R_, C_, S_>(const SplitArgs, const SplitRun, long long);
#define SPLITR_DEFINE(N_, NR_, D_, R_, C_, S_) \
  template __global__ void k_split_reg<N_, NR_, D_, R_, C_, S_>(const SplitArgs, const SplitRun, long long);
