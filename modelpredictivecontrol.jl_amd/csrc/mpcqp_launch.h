// mpcqp_launch.h -- host-side launch entry points of the kernels in mpcqp_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <string>

#include "mpcqp_types.h"

namespace mpcqp {
hipError_t launch_predmat(const Dims& d, const Model& m, bool terminal, hipStream_t st);
hipError_t launch_hessian(const Dims& d, const Model& m, hipStream_t st);
hipError_t launch_step(const Dims& d, const Model& m, const StepIO& io, hipStream_t st);
hipError_t launch_step_spec_or_aot(const Dims& d, const Model& m, const StepIO& io, hipStream_t st);   // never the small-problem kernel
size_t step_lds_bytes(const Dims& d);
// kernel a step runs on: 0 runtime-dimension kernel, 1 ahead-of-time specialisation, 2 on-demand specialisation
int step_kernel_kind(const Dims& d, const Model& m);   // (m: the handle's arrays -- a block / dense M_Hp excludes the small-problem kernel)
int step_kernel_kind_other(const Dims& d);   // ... of the steps the small-problem kernel (kind 3) does not take
int prepare_step(const Dims& d, const Model& m, std::string* err);     // compile (if needed) + load; returns the kind
int prebuild_step(const Dims& d, std::string* err);    // compile only; -1 on failure
// one-time check of an on-demand kernel against the runtime-dimension kernel (see mpcqp_prepare)
bool spec_verified(const Dims& d);
void mark_spec_verified(const Dims& d);
void reject_spec(const Dims& d);
hipError_t launch_step_generic(const Dims& d, const Model& m, const StepIO& io, hipStream_t st);
hipError_t launch_step_unverified_spec(const Dims& d, const Model& m, const StepIO& io, hipStream_t st);   // self-test only
// Small problems (nZ~ <= 16, box and input-bound rows only): four controllers per wavefront (mpcqp_small_bodies.h)
hipError_t launch_step_small(const Dims& d, const Model& m, const StepIO& io, hipStream_t st);
inline bool small_eligible(const Dims& d, const Model& m, const StepIO& io) {
    static const bool on = [] { const char* e = getenv("MPCQP_SMALL"); return !(e && e[0] == '0'); }();
    return on && d.nZ <= 16 && (d.gmask & ~0xFu) == 0 && d.nw == 0 && !d.dense_w && !m.Mblk && !m.Mfull &&
           !(d.flags & (4u | 8u)) && !io.kf_y0m && !io.kf_predict && !io.q_keep && !io.lam_out &&
           (size_t)(2 * d.nY + 16) * 4 * sizeof(double) <= 64 * 1024;
}
hipError_t launch_kf_correct(const Dims& d, const Model& m, const KfParams& kf, double* xhat0,
                             const double* y0m, const double* d0, hipStream_t st);
hipError_t launch_kf_predict(const Dims& d, const Model& m, double* xhat0, const double* u0,
                             const double* d0, hipStream_t st);
}  // namespace mpcqp
