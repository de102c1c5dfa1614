// rydemu_splitreg.hip - part units of librydemu: the k_split_reg instantiations of ONE register size
// (-DRYD_SPLITR_N=12 | 13 | 14), declared `extern template` in rydemu.hip when that is built with -DRYD_SPLIT_TUS.
// Device code + the kernels' host stubs only; no host logic lives here.  See k_split_reg_inst.hpp.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <type_traits>
#include <utility>

#include "../../include/rydemu.h"

typedef double2 cplx;

#include "dev_common.hpp"
#include "split_types.hpp"
#include "k_split_reg.hpp"
#include "k_split_reg_inst.hpp"

#ifndef RYD_SPLITR_N
#error "rydemu_splitreg.hip: -DRYD_SPLITR_N=12, 13 or 14"
#endif
#if RYD_SPLITR_N == 12
SPLITR_INSTANCES_12(SPLITR_DEFINE)
#elif RYD_SPLITR_N == 13
SPLITR_INSTANCES_13(SPLITR_DEFINE)
#elif RYD_SPLITR_N == 14
SPLITR_INSTANCES_14(SPLITR_DEFINE)
#else
#error "rydemu_splitreg.hip: RYD_SPLITR_N must be 12, 13 or 14"
#endif
