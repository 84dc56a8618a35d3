// mhe_launch.h -- host-side launch entry points of the kernels in mhe_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "mhe_types.h"

namespace mpcqp {
namespace mhe {
hipError_t launch_setup(const Dims& d, const Raw& in, double* cst, hipStream_t st);
// mode: bit 0 KalmanFilter correction, bit 1 prediction, bit 2 load P from P0 (ABI layout) first
hipError_t launch_cov(const Dims& d, const Args& a, int mode, const double* P0, double* Pout, hipStream_t st);
hipError_t launch_step(const Dims& d, const Args& a, hipStream_t st);
int waves_for(int device, int B, int NX);      // size of the persistent grid
}  // namespace mhe
}  // namespace mpcqp
