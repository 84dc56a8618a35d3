// mhe_kernels.hip -- gfx950 kernels of the batched linear MovingHorizonEstimator (bodies: mhe_bodies.h).
// One wavefront per workgroup, four estimators per wavefront (one per DPP row), persistent grid.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "mhe_bodies.h"
#include "mhe_devwave.h"
#include "mhe_launch.h"
#include "mpcqp_launch.h"

namespace mpcqp {
namespace mhe {

template <int NX>
__global__ __launch_bounds__(64) void k_mhe_setup(Dims d, Raw in, double* cst) {
    MheDevWave w{(int)threadIdx.x};
    setup_body<MheDevWave, NX>(w, d, in, cst, (int)blockIdx.x);
}
template <int NX>
__global__ __launch_bounds__(64) void k_mhe_cov(Dims d, Args a, int mode, const double* P0, double* Pout) {
    MheDevWave w{(int)threadIdx.x};
    cov_body<MheDevWave, NX>(w, d, a, mode, P0, Pout, (int)blockIdx.x);
}
#ifndef MPCQP_MHE_WAVES
#define MPCQP_MHE_WAVES 2       // register budget of the step kernel, in waves per SIMD
#endif
#ifndef MPCQP_MHE_WAVES_ALL
#define MPCQP_MHE_WAVES_ALL 1   // the same for the all-class / soft variants (CM = 7, 15): one wave per SIMD, 512 registers
#endif
// CM = 1: x̂ bounds only (or none) -- the common case, without the code and registers of the ŵ / v̂ rows
template <int NX, unsigned CM>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(CM == 1u ? MPCQP_MHE_WAVES : MPCQP_MHE_WAVES_ALL, 8))) void k_mhe_step(Dims d, Args a) {
    MheDevWave w{(int)threadIdx.x};
    step_body<MheDevWave, NX, CM>(w, d, a, (int)blockIdx.x, mpcqp_smem);
}

#define MHE_DISPATCH(NXV, CALL)                 \
    switch (NXV) {                              \
        case 4: { constexpr int NX = 4; CALL; } break;   \
        case 8: { constexpr int NX = 8; CALL; } break;   \
        case 12: { constexpr int NX = 12; CALL; } break; \
        case 16: { constexpr int NX = 16; CALL; } break; \
        default: return hipErrorInvalidValue;   \
    }

hipError_t launch_setup(const Dims& d, const Raw& in, double* cst, hipStream_t st) {
    MHE_DISPATCH(d.NX, hipLaunchKernelGGL(k_mhe_setup<NX>, dim3(d.nwaves), dim3(WAVE), 0, st, d, in, cst));
    return hipGetLastError();
}
hipError_t launch_cov(const Dims& d, const Args& a, int mode, const double* P0, double* Pout, hipStream_t st) {
    MHE_DISPATCH(d.NX, hipLaunchKernelGGL(k_mhe_cov<NX>, dim3(d.nwaves), dim3(WAVE), 0, st, d, a, mode, P0, Pout));
    return hipGetLastError();
}
hipError_t launch_step(const Dims& d, const Args& a, hipStream_t st) {
    const size_t lds = step_lds_doubles(d.NX) * sizeof(double);
    if ((d.cls & ~CLS_X) == 0) {
        MHE_DISPATCH(d.NX, hipLaunchKernelGGL((k_mhe_step<NX, 1u>), dim3(d.nwaves), dim3(WAVE), lds, st, d, a));
    } else if (d.cls & CLS_S) {       // soft constraints: all classes + the slack variable
        MHE_DISPATCH(d.NX, hipLaunchKernelGGL((k_mhe_step<NX, 15u>), dim3(d.nwaves), dim3(WAVE), lds, st, d, a));
    } else {
        MHE_DISPATCH(d.NX, hipLaunchKernelGGL((k_mhe_step<NX, 7u>), dim3(d.nwaves), dim3(WAVE), lds, st, d, a));
    }
    return hipGetLastError();
}

}  // namespace mhe

namespace mhe {

int waves_for(int device, int B, int NX) {
    (void)NX;
    int cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    int per_cu = MPCQP_MHE_BMID_LDS ? 8 : 12;      // LDS: 160 KB / (2 or 3 matrices x NX x 512 B)
    if (const char* e = getenv("MPCQP_MHE_WAVES_PER_CU")) per_cu = atoi(e) > 0 ? atoi(e) : per_cu;
    const int groups = (B + GPW - 1) / GPW;
    const int cap = cus * per_cu;
    return groups < cap ? groups : cap;
}

}  // namespace mhe
}  // namespace mpcqp
