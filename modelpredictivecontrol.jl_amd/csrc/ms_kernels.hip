// ms_kernels.hip -- gfx950 kernels of the MultipleShooting LinMPC step (ms_bodies.h): one controller per 64-lane
// wavefront, one wavefront per workgroup.  Two placements of the horizon-long data (iterate, Riccati factor, rows):
//   k_ms_step    everything in LDS (small problems: up to MPCQP_MS_LDS_SHARE bytes per wavefront), grid = B;
//   k_ms_step_g  the model and the work matrices of a stage in LDS, the horizon-long data in a per-wavefront scratch in
//                HBM (L2-resident for the stage in flight); persistent grid of `nslots` wavefronts looping over the batch,
//                so the scratch is nslots x big doubles whatever B -- any horizon, eight wavefronts per CU.
#include <map>
#include <mutex>
#include <utility>
#include <hip/hip_runtime.h>

#include "mpcqp_bodies.h"
#include "mpcqp_devwave.h"
#include "ms_bodies.h"
#include "ms_launch.h"

namespace mpcqp {

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 8))) void k_ms_step(Dims d, Model m, StepIO io, MsIO ms) {
    DevWave w{(int)threadIdx.x};
    ms_step_body<false>(w, d, m, io, ms, (int)blockIdx.x, mpcqp_smem, (double*)nullptr);
}

#ifndef MPCQP_MS_WAVES
#define MPCQP_MS_WAVES 2
#endif
// (two wavefronts per SIMD: without the attribute the compiler takes 372 registers and leaves one)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MPCQP_MS_WAVES, 8))) void k_ms_step_g(Dims d, Model m, StepIO io, MsIO ms, size_t big) {
    DevWave w{(int)threadIdx.x};
    double* scratch = ms.scratch + (size_t)blockIdx.x * big;
    // controllers are handed out one at a time (ms.next: a counter in HBM, zeroed before the launch): a solve that runs to
    // its iteration limit (80 iterations against a mean of 13) then delays its own wavefront, not a fixed share of the
    // batch -- with the static assignment b = blockIdx + k gridDim one such controller set the time of the whole launch
    // (measured: 131 ms for 2048 C3 controllers of which one is at the limit, 25 ms of work per wavefront otherwise)
#ifdef MPCQP_MS_CONST_EXP
    // (experiment: what compile-time dimensions would buy this kernel -- the bench shape 6,2,2,50,50 with its dimensions as constants)
    if (d.nxh == 8 && d.nu == 2 && d.ny == 2 && d.Hp == 50 && d.Hc == 50 && d.neps == 1 && d.nd == 0 && d.nw == 0 && d.default_nb == 1) {
        Dims dc = d;
        dc.nxh = 8; dc.nu = 2; dc.ny = 2; dc.Hp = 50; dc.Hc = 50; dc.neps = 1; dc.nd = 0; dc.nD = 0; dc.nw = 0; dc.nW = 0;
        dc.nDU = 100; dc.nZ = 101; dc.nU = 100; dc.nY = 100; dc.default_nb = 1;
        for (;;) {
            int b = 0;
            if (threadIdx.x == 0) b = atomicAdd(ms.next, 1);
            b = __builtin_amdgcn_readfirstlane(b);
            if (b >= dc.B) break;
            ms_step_body<true>(w, dc, m, io, ms, b, mpcqp_smem, scratch);
            w.sync();
        }
        return;
    }
#endif
    for (;;) {
        int b = 0;
        if (threadIdx.x == 0) b = atomicAdd(ms.next, 1);
        b = __builtin_amdgcn_readfirstlane(b);
        if (b >= d.B) break;
        ms_step_body<true>(w, d, m, io, ms, b, mpcqp_smem, scratch);
        w.sync();
    }
}

size_t ms_lds_bytes(const Dims& d, const Model& m) { return (size_t)make_ms_carve(d, m).total * sizeof(double); }

// bytes of HBM scratch the step needs (0: everything lives in LDS) and the number of resident wavefronts
size_t ms_scratch_bytes(const Dims& d, const Model& m, int* nslots) {
    const MsCarve c = make_ms_carve(d, m);
    if (c.big_in_lds) { if (nslots) *nslots = 0; return 0; }
    const size_t lds = (size_t)c.small * sizeof(double);
    // resident wavefronts per CU: what the runtime says for this kernel and this much LDS (registers, LDS allocation
    // granularity).  A persistent grid larger than that runs its surplus workgroups as a SECOND round (measured: 2048
    // launched where 1792 fit cost 2x).  The answer depends on (device, LDS bytes) only: asked once, then remembered,
    // so that a step does no runtime queries.
    int dev = 0, cus = 256, per_cu = 0;
    (void)hipGetDevice(&dev);
    {
        static std::mutex mu;
        static std::map<std::pair<int, size_t>, std::pair<int, int>> seen;
        std::lock_guard<std::mutex> lock(mu);
        auto it = seen.find({dev, lds});
        if (it == seen.end()) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
            if (ensure_lds((const void*)k_ms_step_g, lds) != hipSuccess ||
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k_ms_step_g, WAVE, lds) != hipSuccess || per_cu < 1) {
                per_cu = (int)((160 * 1024) / (lds ? lds : 1));
                per_cu = per_cu > 8 ? 8 : per_cu < 1 ? 1 : per_cu;
            }
            it = seen.emplace(std::make_pair(dev, lds), std::make_pair(cus, per_cu)).first;
        }
        cus = it->second.first; per_cu = it->second.second;
    }
    int n = cus * per_cu;
    if (n > d.B) n = d.B;
    if (nslots) *nslots = n;
    return (size_t)n * c.big * sizeof(double);
}

hipError_t launch_ms_step(const Dims& d, const Model& m, const StepIO& io, const MsIO& ms, hipStream_t st) {
    const MsCarve c = make_ms_carve(d, m);
    const size_t lds = (size_t)c.total * sizeof(double);
    if (c.big_in_lds) {
        hipError_t e = ensure_lds((const void*)k_ms_step, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_ms_step, dim3(d.B), dim3(WAVE), lds, st, d, m, io, ms);
    } else {
        if (!ms.scratch || ms.nslots < 1) return hipErrorInvalidValue;
        hipError_t e = ensure_lds((const void*)k_ms_step_g, lds);
        if (e != hipSuccess) return e;
        if (!ms.next) return hipErrorInvalidValue;
        e = hipMemsetAsync(ms.next, 0, sizeof(int), st);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_ms_step_g, dim3(ms.nslots), dim3(WAVE), lds, st, d, m, io, ms, (size_t)c.big);
    }
    return hipGetLastError();
}

}  // namespace mpcqp
