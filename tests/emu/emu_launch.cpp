// TEST INFRASTRUCTURE ONLY: runs the kernel bodies of csrc/mpcqp_bodies.h on the CPU, one host
// fiber per lane of a 64-wide "wavefront" (emu_fiber.h), problems one after the other.  See fakehip/.
#include <barrier>
#include <thread>
#include <vector>

#include <cstdlib>

#include "mpcqp_bodies.h"
#include "mpcqp_dispatch.h"
#include "mpcqp_launch.h"

// specialisations of the small test problems of tests/test_abi_and_host.py, so that the
// compile-time-dims code path (incl. move-blocking vectors and nd > 0) is exercised on the CPU too
#define MPCQP_EMU_SPECIALIZATIONS(X) \
    MPCQP_SPECIALIZATIONS(X)         \
    XNB(1, 1, 3, 6, 3, 1, 0x09Fu)    \
    XNB(1, 1, 3, 6, 3, 1, 0x39Fu)    \
    XNB(1, 1, 3, 6, 3, 1, 0x08Cu)

#define MPCQP_EMU_FIBER_IMPL        // (the context switch of emu_fiber.h is assembled in this unit)
#include "emu_wave.h"

namespace mpcqp {

hipError_t launch_predmat(const Dims& d, const Model& m, bool terminal, hipStream_t) {
    run_waves(d.B, predmat_lds_doubles(d), [&](EmuWave& w, int b, double* sm) { predmat_body(w, d, m, b, sm, terminal); });
    return hipSuccess;
}
hipError_t launch_hessian(const Dims& d, const Model& m, hipStream_t) {
    const char* fg = getenv("MPCQP_FORCE_GENERIC");
    if (!(fg && fg[0] == '1') && !m.Mfull && !m.Ndense && !m.Ldense) {
#define XNB(NU, NY, NXH, HP, HC, NEPS, GM) XX(NU, NY, NXH, HP, HC, NEPS, GM, 0)
#define X(NU, NY, NXH, HP, HC, NEPS, GM) XX(NU, NY, NXH, HP, HC, NEPS, GM, 1)
#define XX(NU, NY, NXH, HP, HC, NEPS, GM, NB)                                                  \
        {                                                                                      \
            using SD = StaticDims<NU, NY, NXH, HP, HC, NEPS, GM, NB>;                          \
            if (SD::matches_dims(d)) {                                                         \
                const SD sd(d);                                                                \
                run_waves(d.B, make_carve(sd).total, [&](EmuWave& w, int b, double* sm) {      \
                    hessian_body(w, sd, m, b, sm); });                                         \
                return hipSuccess;                                                             \
            }                                                                                  \
        }
        MPCQP_EMU_SPECIALIZATIONS(X)
#undef X
#undef XX
#undef XNB
    }
    run_waves(d.B, make_carve(d).total, [&](EmuWave& w, int b, double* sm) { hessian_body(w, d, m, b, sm); });
    return hipSuccess;
}
hipError_t launch_step(const Dims& d, const Model& m, const StepIO& io, hipStream_t st) {
    const char* fg = getenv("MPCQP_FORCE_GENERIC");
    if (!(fg && fg[0] == '1') && small_eligible(d, m, io)) return launch_step_small(d, m, io, st);
    return launch_step_spec_or_aot(d, m, io, st);
}
hipError_t launch_step_spec_or_aot(const Dims& d, const Model& m, const StepIO& io, hipStream_t) {
    const char* fg = getenv("MPCQP_FORCE_GENERIC");
    if (!(fg && fg[0] == '1') && !d.dense_w) {
#define XNB(NU, NY, NXH, HP, HC, NEPS, GM) XX(NU, NY, NXH, HP, HC, NEPS, GM, 0)
#define X(NU, NY, NXH, HP, HC, NEPS, GM) XX(NU, NY, NXH, HP, HC, NEPS, GM, 1)
#define XX(NU, NY, NXH, HP, HC, NEPS, GM, NB)                                                  \
        {                                                                                      \
            using SD = StaticDims<NU, NY, NXH, HP, HC, NEPS, GM, NB>;                          \
            if (SD::matches(d)) {                                                              \
                const SD sd(d);                                                                \
                run_waves(d.B, make_carve(sd).total, [&](EmuWave& w, int b, double* sm) {      \
                    step_body(w, sd, m, io, b, sm); });                                        \
                return hipSuccess;                                                             \
            }                                                                                  \
        }
        MPCQP_EMU_SPECIALIZATIONS(X)
#undef X
#undef XX
#undef XNB
    }
    run_waves(d.B, make_carve(d).total, [&](EmuWave& w, int b, double* sm) { step_body(w, d, m, io, b, sm); });
    return hipSuccess;
}
hipError_t launch_kf_correct(const Dims& d, const Model& m, const KfParams& kf, double* xhat0,
                             const double* y0m, const double* d0, hipStream_t) {
    std::vector<double> xin(xhat0, xhat0 + (size_t)d.B * d.nxh);     // lanes are not in lockstep here
    for (int b = 0; b < d.B; ++b)
        for (int i = 0; i < d.nxh; ++i) kf_correct_lane(d, m, kf, b, i, xin.data(), xhat0, y0m, d0);
    return hipSuccess;
}
hipError_t launch_kf_predict(const Dims& d, const Model& m, double* xhat0, const double* u0,
                             const double* d0, hipStream_t) {
    std::vector<double> xin(xhat0, xhat0 + (size_t)d.B * d.nxh);
    for (int b = 0; b < d.B; ++b)
        for (int i = 0; i < d.nxh; ++i) kf_predict_lane(d, m, b, i, xin.data(), xhat0, u0, d0);
    return hipSuccess;
}
// (the emulator has no on-demand kernels: every shape it was not compiled for runs the runtime-dims body)
int step_kernel_kind(const Dims& d, const Model& m) { return small_eligible(d, m, StepIO{}) ? 3 : 0; }
int step_kernel_kind_other(const Dims&) { return 0; }
int prepare_step(const Dims& d, const Model& m, std::string*) { return small_eligible(d, m, StepIO{}) ? 3 : 0; }
int prebuild_step(const Dims&, std::string*) { return 0; }
bool spec_verified(const Dims&) { return true; }
bool spec_present(const Dims&) { return false; }
void mark_spec_verified(const Dims&) {}
void reject_spec(const Dims&) {}
hipError_t launch_step_generic(const Dims& d, const Model& m, const StepIO& io, hipStream_t st) { return launch_step(d, m, io, st); }
hipError_t launch_step_unverified_spec(const Dims& d, const Model& m, const StepIO& io, hipStream_t st) { return launch_step(d, m, io, st); }
size_t step_lds_bytes(const Dims& d) { return (size_t)make_carve(d).total * sizeof(double); }

}  // namespace mpcqp
