// mhe_devwave.h -- gfx950 wave interface of the MovingHorizonEstimator kernels: DevWave plus the
// 16-lane-row primitives (one estimator per DPP row).
#pragma once
#include "mpcqp_devwave.h"

namespace mpcqp {
namespace mhe {

struct MheDevWave : DevWave {
    // value of v in lane C of this lane's 16-lane row: one `v_mov_b64_dpp ... row_newbcast:C`
    // (gfx90a+: the only DPP control the 64-bit move supports)
    template <int C>
    __device__ __forceinline__ double rowbc(double v) const {
        return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + C, 0xf, 0xf, true);
    }
    template <class Op>
    static __device__ __forceinline__ double rowred(double v, Op op) {
        v = op(v, dpp<0xB1>(v));     // quad_perm [1,0,3,2]
        v = op(v, dpp<0x4E>(v));     // quad_perm [2,3,0,1]
        v = op(v, dpp<0x141>(v));    // row_half_mirror
        v = op(v, dpp<0x140>(v));    // row_mirror
        return v;
    }
    __device__ __forceinline__ double rsum(double v) const { return rowred(v, [](double x, double y) { return x + y; }); }
    __device__ __forceinline__ double rmin(double v) const { return rowred(v, [](double x, double y) { return fmin(x, y); }); }
    __device__ __forceinline__ double rmax(double v) const { return rowred(v, [](double x, double y) { return fmax(x, y); }); }
};

}  // namespace mhe
}  // namespace mpcqp
