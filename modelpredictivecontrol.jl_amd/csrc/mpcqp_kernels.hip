// mpcqp_kernels.hip -- gfx950 kernels of the batched LinMPC step: one QP per 64-lane wavefront,
// one wavefront per workgroup, the condensed problem (step-response table, packed normal-equation
// tile, residual/row state) staged in LDS.  Bodies: mpcqp_bodies.h.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "mpcqp_bodies.h"
#include "mpcqp_dispatch.h"
#include "mpcqp_launch.h"

#ifndef MPCQP_STEP_WAVES
#define MPCQP_STEP_WAVES 2      // register budget of the specialised step kernel, in waves per SIMD
#endif

namespace mpcqp {

// Wave-level primitives without LDS traffic: reductions run on DPP lane permutes inside each
// 16-lane row and v_readlane across the four rows; broadcasts of a wave-uniform lane are two
// v_readlane.  (ds_bpermute-based __shfl costs an LDS round trip per step, and this kernel's
// critical path is a chain of ~120 broadcasts + ~12 reductions per IPM iteration.)
struct DevWave {
    int lane;
    // One wavefront per workgroup: LDS operations of a wave execute in issue order, so ordering
    // LDS traffic between lanes needs no s_barrier and no s_waitcnt -- only a fence the compiler
    // may not move memory operations across.
    __device__ __forceinline__ void sync() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    template <int CTRL>
    static __device__ __forceinline__ double dpp(double v) {
        int lo = __double2loint(v), hi = __double2hiint(v);
        lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
        hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
        return __hiloint2double(hi, lo);
    }
    static __device__ __forceinline__ double lane_value(double v, int src) {   // src wave-uniform
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
        return __hiloint2double(hi, lo);
    }
    template <class Op>
    static __device__ __forceinline__ double reduce(double v, Op op) {
        v = op(v, dpp<0xB1>(v));     // quad_perm [1,0,3,2]
        v = op(v, dpp<0x4E>(v));     // quad_perm [2,3,0,1]
        v = op(v, dpp<0x141>(v));    // row_half_mirror
        v = op(v, dpp<0x140>(v));    // row_mirror: every lane holds its 16-lane row's value
        const double a = lane_value(v, 0), b = lane_value(v, 16);
        const double c = lane_value(v, 32), d = lane_value(v, 48);
        return op(op(a, b), op(c, d));
    }
    // sum over each aligned group of four lanes (result in all four)
    __device__ __forceinline__ double quad_sum(double v) {
        v += dpp<0xB1>(v);
        v += dpp<0x4E>(v);
        return v;
    }
    __device__ __forceinline__ double sum(double v) { return reduce(v, [](double x, double y) { return x + y; }); }
    __device__ __forceinline__ double minv(double v) { return reduce(v, [](double x, double y) { return fmin(x, y); }); }
    __device__ __forceinline__ double maxv(double v) { return reduce(v, [](double x, double y) { return fmax(x, y); }); }
    __device__ __forceinline__ int isum(int v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    }
    __device__ __forceinline__ double bcast(double v, int src) { return lane_value(v, src); }
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
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MPCQP_STEP_WAVES, 8))) void k_step_s(Dims d, Model m, StepIO io) {
    DevWave w{(int)threadIdx.x};
    const SD sd(d);
    step_body(w, sd, m, io, (int)blockIdx.x, mpcqp_smem);
}

// SteadyKalmanFilter steps: npad = next power of two >= nx̂ lanes per problem, 256-thread blocks
__global__ __launch_bounds__(256) void k_kf_correct(Dims d, Model m, KfParams kf, double* xhat0,
                                                    const double* y0m, const double* d0, int npad) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    kf_correct_lane(d, m, kf, g / npad, g % npad, xhat0, xhat0, y0m, d0);
}

__global__ __launch_bounds__(256) void k_kf_predict(Dims d, Model m, double* xhat0, const double* u0,
                                                    const double* d0, int npad) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    kf_predict_lane(d, m, g / npad, g % npad, xhat0, xhat0, u0, d0);
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

static int kf_npad(const Dims& d) {
    int n = 1;
    while (n < d.nxh) n <<= 1;
    return n;
}

hipError_t launch_kf_correct(const Dims& d, const Model& m, const KfParams& kf, double* xhat0,
                             const double* y0m, const double* d0, hipStream_t st) {
    const int npad = kf_npad(d);
    const long long total = (long long)d.B * npad;
    hipLaunchKernelGGL(k_kf_correct, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d, m, kf,
                       xhat0, y0m, d0, npad);
    return hipGetLastError();
}

hipError_t launch_kf_predict(const Dims& d, const Model& m, double* xhat0, const double* u0,
                             const double* d0, hipStream_t st) {
    const int npad = kf_npad(d);
    const long long total = (long long)d.B * npad;
    hipLaunchKernelGGL(k_kf_predict, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d, m,
                       xhat0, u0, d0, npad);
    return hipGetLastError();
}

size_t step_lds_bytes(const Dims& d) { return (size_t)make_carve(d).total * sizeof(double); }

}  // namespace mpcqp
