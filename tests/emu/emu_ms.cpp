// TEST INFRASTRUCTURE ONLY: the MultipleShooting step body (csrc/ms_bodies.h) on the CPU wave emulator.
#include "emu_wave.h"
#include "ms_bodies.h"
#include "ms_launch.h"

namespace mpcqp {
hipError_t launch_ms_step(const Dims& d, const Model& m, const StepIO& io, const MsIO& ms, hipStream_t) {
    const MsCarve c = make_ms_carve(d, m);
    // (both placements of the horizon-long data: behind the small block in "LDS", or in the scratch the host allocated)
    if (c.big_in_lds) run_waves(d.B, c.total, [&](EmuWave& w, int b, double* sm) { ms_step_body<false>(w, d, m, io, ms, b, sm, (double*)nullptr); });
    else run_waves(d.B, c.total, [&](EmuWave& w, int b, double* sm) { ms_step_body<true>(w, d, m, io, ms, b, sm, ms.scratch); });
    return hipSuccess;
}
size_t ms_lds_bytes(const Dims& d, const Model& m) { return (size_t)make_ms_carve(d, m).total * sizeof(double); }
size_t ms_scratch_bytes(const Dims& d, const Model& m, int* nslots) {
    const MsCarve c = make_ms_carve(d, m);
    if (nslots) *nslots = c.big_in_lds ? 0 : 1;
    return c.big_in_lds ? 0 : (size_t)c.big * sizeof(double);
}
}  // namespace mpcqp
