// small_kernels.hip -- gfx950 kernels of the LinMPC step for SMALL problems (nZ~ <= 16: four controllers per wavefront, one
// per 16-lane DPP row; bodies: mpcqp_small_bodies.h).  A translation unit of its own: the variants with output-bound rows
// are twelve register-heavy kernels, and the library's units compile in parallel.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "mhe_bodies.h"
#include "mhe_devwave.h"
#include "mhe_launch.h"
#include "mpcqp_launch.h"
#include "mpcqp_small_bodies.h"

namespace mpcqp {
namespace mhe {
#define MHE_DISPATCH(NXV, CALL)                 \
    switch (NXV) {                              \
        case 4: { constexpr int NX = 4; CALL; } break;   \
        case 8: { constexpr int NX = 8; CALL; } break;   \
        case 12: { constexpr int NX = 12; CALL; } break; \
        case 16: { constexpr int NX = 16; CALL; } break; \
        default: return hipErrorInvalidValue;   \
    }

}  // namespace mhe

#ifndef MPCQP_SMALL_WAVES
#define MPCQP_SMALL_WAVES 2      // register budget of the small-problem step kernel, in waves per SIMD
#endif
template <int NX>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MPCQP_SMALL_WAVES, 8))) void k_step_small(Dims d, Model m, StepIO io) {
    mhe::MheDevWave w{(int)threadIdx.x};
    step_small_body<mhe::MheDevWave, NX, 0, true>(w, d, m, io, (int)blockIdx.x, mpcqp_smem);
}
// the same on a one-wave-per-SIMD register budget (no spills): for grids that leave most of the chip idle anyway -- up to one
// wavefront per SIMD, i.e. B <= 4096 on 256 CUs, where nothing else would hide the scratch round trips of the spilled registers
// (B = 1024: 0.116 -> 0.107 ms, B = 4096: 0.14 -> 0.12 ms; from B = 16384 on the two-wave kernel is 25 % faster)
template <int NX>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 8))) void k_step_small_w1(Dims d, Model m, StepIO io) {
    mhe::MheDevWave w{(int)threadIdx.x};
    step_small_body<mhe::MheDevWave, NX, 0>(w, d, m, io, (int)blockIdx.x, mpcqp_smem);
}
// with output-bound rows: their slacks and multipliers (KYS rows of both sides per lane) want the registers of a
// whole SIMD lane file, and the dense E in LDS bounds the occupancy anyway: one wave per SIMD
template <int NX, int KYS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 8))) void k_step_small_y(Dims d, Model m, StepIO io) {
    mhe::MheDevWave w{(int)threadIdx.x};
    step_small_body<mhe::MheDevWave, NX, KYS>(w, d, m, io, (int)blockIdx.x, mpcqp_smem);
}

// largest grid of the one-wave-per-SIMD variant: one wavefront per SIMD of the device
static unsigned small_w1_grid() {
    static const unsigned n = [] {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            return (unsigned)prop.multiProcessorCount * 4u;
        return 1024u;
    }();
    return n;
}

hipError_t launch_step_small(const Dims& d, const Model& m, const StepIO& io, hipStream_t st) {
    const bool hasy = small_has_y(d);
    const size_t lds = small_lds_doubles(d, hasy) * sizeof(double);
    const unsigned grid = (unsigned)((d.B + SMALL_GPW - 1) / SMALL_GPW);
    const int NXv = 4 * ((d.nZ + 3) / 4);
    if (hasy) {
        switch (small_row_slots(d)) {
            case 2: MHE_DISPATCH(NXv, hipLaunchKernelGGL((k_step_small_y<NX, 2>), dim3(grid), dim3(WAVE), lds, st, d, m, io)); break;
            case 3: MHE_DISPATCH(NXv, hipLaunchKernelGGL((k_step_small_y<NX, 3>), dim3(grid), dim3(WAVE), lds, st, d, m, io)); break;
            default: MHE_DISPATCH(NXv, hipLaunchKernelGGL((k_step_small_y<NX, 4>), dim3(grid), dim3(WAVE), lds, st, d, m, io)); break;
        }
    } else if (grid <= small_w1_grid()) {
        MHE_DISPATCH(NXv, hipLaunchKernelGGL(k_step_small_w1<NX>, dim3(grid), dim3(WAVE), lds, st, d, m, io));
    } else {
        MHE_DISPATCH(NXv, hipLaunchKernelGGL(k_step_small<NX>, dim3(grid), dim3(WAVE), lds, st, d, m, io));
    }
    return hipGetLastError();
}

}  // namespace mpcqp
