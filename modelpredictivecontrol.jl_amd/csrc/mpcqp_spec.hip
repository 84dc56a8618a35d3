// mpcqp_spec.hip -- ONE specialisation of the step / Hessian kernels, compiled on demand by
// libmpcqp.so for dimensions that are not in its ahead-of-time list (mpcqp_dispatch.h):
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -DMPCQP_SPEC_DIMS=NU,NY,NXH,HP,HC,NEPS,GMASK,DEFAULT_NB ...
// The resulting object is dlopen'ed and cached next to the library (see jit_specialise()).
#include <hip/hip_runtime.h>

#include "mpcqp_bodies.h"
#include "mpcqp_devwave.h"

#ifndef MPCQP_SPEC_DIMS
#error "define MPCQP_SPEC_DIMS=NU,NY,NXH,HP,HC,NEPS,GMASK,DEFAULT_NB"
#endif

using SpecDims = mpcqp::StaticDims<MPCQP_SPEC_DIMS>;

extern "C" {
int mpcqp_spec_matches(const mpcqp::Dims* d) { return SpecDims::matches(*d) ? 1 : 0; }
int mpcqp_spec_matches_dims(const mpcqp::Dims* d) { return SpecDims::matches_dims(*d) ? 1 : 0; }
int mpcqp_spec_launch_step(const mpcqp::Dims* d, const mpcqp::Model* m, const mpcqp::StepIO* io, void* st) {
    return (int)mpcqp::launch_step_static<SpecDims>(*d, *m, *io, (hipStream_t)st);
}
int mpcqp_spec_launch_hessian(const mpcqp::Dims* d, const mpcqp::Model* m, void* st) {
    return (int)mpcqp::launch_hessian_static<SpecDims>(*d, *m, (hipStream_t)st);
}
}
