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
bool spec_present(const Dims& d);      // an on-demand object of this shape is loaded / loadable (verified or not)
void mark_spec_verified(const Dims& d);
void reject_spec(const Dims& d);
hipError_t launch_step_generic(const Dims& d, const Model& m, const StepIO& io, hipStream_t st);
hipError_t launch_step_unverified_spec(const Dims& d, const Model& m, const StepIO& io, hipStream_t st);   // self-test only
// Small problems (nZ~ <= 16; box, input-bound and -- nY <= 64 -- output-bound rows): four controllers per wavefront
// (mpcqp_small_bodies.h)
hipError_t launch_step_small(const Dims& d, const Model& m, const StepIO& io, hipStream_t st);
inline int small_dense_rows_(const Dims& d) {         // = small_dense_rows(d) of mpcqp_small_bodies.h: output-bound + terminal rows
    return (((d.gmask >> (2 * P_Y)) & 3u) ? d.nY : 0) + (((d.gmask >> (2 * P_X)) & 3u) ? d.nxh : 0);
}
inline bool small_has_y(const Dims& d) { return small_dense_rows_(d) > 0; }       // the variant with dense rows (k_step_small_y)
inline size_t small_lds_bytes(const Dims& d) {       // = small_lds_doubles(d, small_has_y(d)) * 8 (mpcqp_small_bodies.h)
    const int NXv = 4 * ((d.nZ + 3) / 4), nR = small_dense_rows_(d);
    const int kys = nR <= 32 ? 2 : nR <= 48 ? 3 : 4;
    return (size_t)4 * ((size_t)(2 * d.nY + 16) + (nR ? (size_t)nR * NXv + 2 * nR + 64 * kys + d.nxh : 0)) * sizeof(double);
}
inline bool small_eligible(const Dims& d, const Model& m, const StepIO& io) {
    static const bool on = [] { const char* e = getenv("MPCQP_SMALL"); return !(e && e[0] == '0'); }();
    static const bool ony = [] { const char* e = getenv("MPCQP_SMALL_Y"); return !(e && e[0] == '0'); }();
    const uint32_t groups = 0xFu | (3u << (2 * P_Y)) | (3u << (2 * P_X));      // box, U, Y, terminal
    const bool termok = !((d.gmask >> (2 * P_X)) & 3u) || m.exT;               // (terminal tables built by K1)
    return on && d.nZ <= 16 && (d.gmask & ~groups) == 0 && (!small_has_y(d) || (ony && small_dense_rows_(d) <= 64 && termok)) &&
           d.nw == 0 && !d.dense_w && !m.Mblk && !m.Mfull && !(d.flags & (4u | 8u)) && !io.kf_y0m && !io.kf_predict && !io.q_keep &&
           !io.lam_out && small_lds_bytes(d) <= 64 * 1024;
}
hipError_t launch_kf_correct(const Dims& d, const Model& m, const KfParams& kf, double* xhat0,
                             const double* y0m, const double* d0, hipStream_t st);
hipError_t launch_kf_predict(const Dims& d, const Model& m, double* xhat0, const double* u0,
                             const double* d0, hipStream_t st);
}  // namespace mpcqp
