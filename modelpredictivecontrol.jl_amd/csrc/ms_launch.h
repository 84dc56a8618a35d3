// ms_launch.h -- host-side launch entry points of the MultipleShooting step kernel (ms_kernels.hip; tests/emu/emu_ms.cpp).
#pragma once
#include <hip/hip_runtime.h>

#include "mpcqp_types.h"

namespace mpcqp {
struct MsIO;
hipError_t launch_ms_step(const Dims& d, const Model& m, const StepIO& io, const MsIO& ms, hipStream_t st);
size_t ms_lds_bytes(const Dims& d, const Model& m);
size_t ms_scratch_bytes(const Dims& d, const Model& m, int* nslots);    // HBM scratch of the step (0: all in LDS)
}  // namespace mpcqp
