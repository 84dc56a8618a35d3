// TEST INFRASTRUCTURE ONLY: the MultipleShooting step body (csrc/ms_bodies.h) on the CPU wave emulator.
#include "emu_wave.h"
#include "ms_bodies.h"
#include "ms_launch.h"

namespace mpcqp {
hipError_t launch_ms_step(const Dims& d, const Model& m, const StepIO& io, const MsIO& ms, hipStream_t) {
    run_waves(d.B, make_ms_carve(d, m).total, [&](EmuWave& w, int b, double* sm) { ms_step_body(w, d, m, io, ms, b, sm); });
    return hipSuccess;
}
size_t ms_lds_bytes(const Dims& d, const Model& m) { return (size_t)make_ms_carve(d, m).total * sizeof(double); }
}  // namespace mpcqp
