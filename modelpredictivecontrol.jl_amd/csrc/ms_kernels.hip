// ms_kernels.hip -- gfx950 kernel of the MultipleShooting LinMPC step (ms_bodies.h): one controller per 64-lane
// wavefront, one wavefront per workgroup, the stage data (iterate, Riccati factor, rows) in LDS.
#include <hip/hip_runtime.h>

#include "mpcqp_bodies.h"
#include "mpcqp_devwave.h"
#include "ms_bodies.h"
#include "ms_launch.h"

namespace mpcqp {

__global__ __launch_bounds__(64) void k_ms_step(Dims d, Model m, StepIO io, MsIO ms) {
    DevWave w{(int)threadIdx.x};
    ms_step_body(w, d, m, io, ms, (int)blockIdx.x, mpcqp_smem);
}

size_t ms_lds_bytes(const Dims& d, const Model& m) { return (size_t)make_ms_carve(d, m).total * sizeof(double); }

hipError_t launch_ms_step(const Dims& d, const Model& m, const StepIO& io, const MsIO& ms, hipStream_t st) {
    const size_t lds = ms_lds_bytes(d, m);
    hipError_t e = ensure_lds((const void*)k_ms_step, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_ms_step, dim3(d.B), dim3(WAVE), lds, st, d, m, io, ms);
    return hipGetLastError();
}

}  // namespace mpcqp
