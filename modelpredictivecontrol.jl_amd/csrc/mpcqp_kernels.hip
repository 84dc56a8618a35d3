// mpcqp_kernels.hip -- gfx950 kernels of the batched LinMPC step: one QP per 64-lane wavefront,
// one wavefront per workgroup, the condensed problem (step-response table, packed normal-equation
// tile, residual/row state) staged in LDS.  Bodies: mpcqp_bodies.h.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "mpcqp_bodies.h"
#include "mpcqp_dispatch.h"
#include "mpcqp_launch.h"

namespace mpcqp {

struct DevWave {
    int lane;
    __device__ __forceinline__ void sync() { __syncthreads(); }
    __device__ __forceinline__ double sum(double v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    }
    __device__ __forceinline__ double minv(double v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
        return v;
    }
    __device__ __forceinline__ double maxv(double v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
        return v;
    }
    __device__ __forceinline__ int isum(int v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    }
    __device__ __forceinline__ double bcast(double v, int src) { return __shfl(v, src, 64); }
};

extern __shared__ __attribute__((aligned(16))) double mpcqp_smem[];

__global__ __launch_bounds__(64) void k_predmat(Dims d, Model m, int terminal) {
    DevWave w{(int)threadIdx.x};
    predmat_body(w, d, m, (int)blockIdx.x, mpcqp_smem, terminal != 0);
}

__global__ __launch_bounds__(64) void k_hessian(Dims d, Model m) {
    DevWave w{(int)threadIdx.x};
    hessian_body(w, d, m, (int)blockIdx.x, mpcqp_smem);
}

__global__ __launch_bounds__(64) void k_step(Dims d, Model m, StepIO io) {
    DevWave w{(int)threadIdx.x};
    step_body(w, d, m, io, (int)blockIdx.x, mpcqp_smem);
}

template <class SD>
__global__ __launch_bounds__(64) void k_hessian_s(Dims d, Model m) {
    DevWave w{(int)threadIdx.x};
    const SD sd(d);
    hessian_body(w, sd, m, (int)blockIdx.x, mpcqp_smem);
}

// specialised on compile-time dimensions (mpcqp_dispatch.h)
template <class SD>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 8))) void k_step_s(Dims d, Model m, StepIO io) {
    DevWave w{(int)threadIdx.x};
    const SD sd(d);
    step_body(w, sd, m, io, (int)blockIdx.x, mpcqp_smem);
}

// ---- launchers (host) ------------------------------------------------------------------------
static hipError_t ensure_lds(const void* fn, size_t bytes) {
    if (bytes <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

hipError_t launch_predmat(const Dims& d, const Model& m, bool terminal, hipStream_t st) {
    size_t lds = (size_t)predmat_lds_doubles(d) * sizeof(double);
    hipError_t e = ensure_lds((const void*)k_predmat, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_predmat, dim3(d.B), dim3(WAVE), lds, st, d, m, terminal ? 1 : 0);
    return hipGetLastError();
}

static bool force_generic();

hipError_t launch_hessian(const Dims& d, const Model& m, hipStream_t st) {
    if (!force_generic()) {
#define X(NU, NY, NXH, HP, HC, NEPS, GM)                                                        \
        {                                                                                       \
            using SD = StaticDims<NU, NY, NXH, HP, HC, NEPS, GM>;                               \
            if (SD::matches_dims(d)) {                                                          \
                const size_t lds_s = (size_t)make_carve(SD(d)).total * sizeof(double);          \
                hipError_t e = ensure_lds((const void*)k_hessian_s<SD>, lds_s);                 \
                if (e != hipSuccess) return e;                                                  \
                hipLaunchKernelGGL(k_hessian_s<SD>, dim3(d.B), dim3(WAVE), lds_s, st, d, m);    \
                return hipGetLastError();                                                       \
            }                                                                                   \
        }
        MPCQP_SPECIALIZATIONS(X)
#undef X
    }
    size_t lds = (size_t)make_carve(d).total * sizeof(double);
    hipError_t e = ensure_lds((const void*)k_hessian, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_hessian, dim3(d.B), dim3(WAVE), lds, st, d, m);
    return hipGetLastError();
}

static bool force_generic() {
    static const bool f = [] { const char* e = getenv("MPCQP_FORCE_GENERIC"); return e && e[0] == '1'; }();
    return f;
}

hipError_t launch_step(const Dims& d, const Model& m, const StepIO& io, hipStream_t st) {
    if (!force_generic()) {
#define X(NU, NY, NXH, HP, HC, NEPS, GM)                                                        \
        {                                                                                       \
            using SD = StaticDims<NU, NY, NXH, HP, HC, NEPS, GM>;                               \
            if (SD::matches(d)) {                                                               \
                const size_t lds_s = (size_t)make_carve(SD(d)).total * sizeof(double);          \
                hipError_t e = ensure_lds((const void*)k_step_s<SD>, lds_s);                    \
                if (e != hipSuccess) return e;                                                  \
                hipLaunchKernelGGL(k_step_s<SD>, dim3(d.B), dim3(WAVE), lds_s, st, d, m, io);   \
                return hipGetLastError();                                                       \
            }                                                                                   \
        }
        MPCQP_SPECIALIZATIONS(X)
#undef X
    }
    size_t lds = (size_t)make_carve(d).total * sizeof(double);
    hipError_t e = ensure_lds((const void*)k_step, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_step, dim3(d.B), dim3(WAVE), lds, st, d, m, io);
    return hipGetLastError();
}

size_t step_lds_bytes(const Dims& d) { return (size_t)make_carve(d).total * sizeof(double); }

}  // namespace mpcqp
