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
    // a wave-uniform pointer, pinned to scalar registers: keeps loop strength reduction from merging it with
    // a lane offset into a per-lane 64-bit pointer that lives in two VGPRs for the whole loop
    template <class T>
    __device__ __forceinline__ T* uniform(T* p) const {
        const uint64_t v = (uint64_t)p;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return (T*)(((uint64_t)hi << 32) | lo);
    }
    // raw buffer access to a wave-private array: scalar (slot) offset + one per-lane byte offset; a lane whose
    // offset is BUF_OOB reads zero and its stores are dropped by the bounds check of the buffer resource
    using Buf = __amdgpu_buffer_rsrc_t;
    static constexpr unsigned BUF_OOB = 0xFFFFFFF0u;
    __device__ __forceinline__ Buf make_buf(double* base, size_t bytes) const {
        return __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes, 0x00020000);
    }
    __device__ __forceinline__ double bload(Buf b, unsigned voff, int soff) const {
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(b, voff, soff, 0));
    }
    __device__ __forceinline__ void bstore(Buf b, unsigned voff, int soff, double v) const {
        using V = decltype(__builtin_amdgcn_raw_buffer_load_b64(b, 0u, 0, 0));
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(V, v), b, voff, soff, 0);
    }
    // acc += sum_i (x_i of lane L_i of this lane's row) * y_i: the broadcast is the DPP modifier of the
    // multiply-add itself.  One asm block per four terms: the hardware does not interlock a DPP read of a
    // VGPR that a VALU instruction wrote within the previous two wait states, and the compiler's hazard
    // recogniser does not see inside inline asm -- the leading s_nop covers every source of the block (none
    // of them is written inside it).
    template <int L0, int L1, int L2, int L3>
    __device__ __forceinline__ void fmabc4(double& acc, double x0, double x1, double x2, double x3, double y0, double y1,
                                           double y2, double y3) const {
        asm("s_nop 1\n\t"
            "v_fmac_f64_dpp %0, %1, %5 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %0, %2, %6 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %0, %3, %7 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %0, %4, %8 row_newbcast:%12 row_mask:0xf bank_mask:0xf"
            : "+v"(acc)
            : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(y0), "v"(y1), "v"(y2), "v"(y3), "n"(L0), "n"(L1), "n"(L2), "n"(L3));
    }
    // a_i += (x of lane L_i of this lane's row) * y_i, four accumulators: a row of a rank-one update whose left factor is
    // spread over the lanes (one v_fmac_f64_dpp per element instead of a broadcast move, selects and an add)
    template <int L0, int L1, int L2, int L3>
    __device__ __forceinline__ void rank1bc4(double& a0, double& a1, double& a2, double& a3, double x, double y0, double y1,
                                             double y2, double y3) const {
        asm("s_nop 1\n\t"
            "v_fmac_f64_dpp %0, %4, %5 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %4, %6 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %2, %4, %7 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %3, %4, %8 row_newbcast:%12 row_mask:0xf bank_mask:0xf"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
            : "v"(x), "v"(y0), "v"(y1), "v"(y2), "v"(y3), "n"(L0), "n"(L1), "n"(L2), "n"(L3));
    }
    // Gauss-Jordan row update of four elements: a_i <- m a_i + g (a_i of lane K)
    template <int K>
    __device__ __forceinline__ void gjrow4(double& a0, double& a1, double& a2, double& a3, double m, double g) const {
        double t0, t1, t2, t3;
        asm("v_mul_f64 %0, %4, %8\n\t"
            "v_mul_f64 %1, %5, %8\n\t"
            "v_mul_f64 %2, %6, %8\n\t"
            "v_mul_f64 %3, %7, %8\n\t"
            "v_fmac_f64_dpp %0, %4, %9 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %5, %9 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %2, %6, %9 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %3, %7, %9 row_newbcast:%10 row_mask:0xf bank_mask:0xf"
            : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
            : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(m), "v"(g), "n"(K));
        a0 = t0; a1 = t1; a2 = t2; a3 = t3;
    }
    // acc -= sum_i (x_i of lane L_i of this lane's row) * y_i (the negation is the source modifier of the multiply-add)
    template <int L0, int L1, int L2, int L3>
    __device__ __forceinline__ void fmsbc4(double& acc, double x0, double x1, double x2, double x3, double y0, double y1,
                                           double y2, double y3) const {
        asm("s_nop 1\n\t"
            "v_fmac_f64_dpp %0, %1, -%5 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %0, %2, -%6 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %0, %3, -%7 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %0, %4, -%8 row_newbcast:%12 row_mask:0xf bank_mask:0xf"
            : "+v"(acc)
            : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(y0), "v"(y1), "v"(y2), "v"(y3), "n"(L0), "n"(L1), "n"(L2), "n"(L3));
    }
    // Gauss-Jordan elimination step on four elements, in place: a_i <- a_i + g (a_i of lane K); the pivot lane passes g = 0
    // and keeps its row (Ops::gj scales the rows once, after the last pivot).  The trailing s_nop: the next DPP read of a result
    // may be the compiler's (rowbc of the next pivot), and its hazard recogniser does not see the writes inside the block.
    // (A pivot broadcast written as asm -- s_nop + v_mov_b64_dpp without bound_ctrl -- made k_step_small_y<8,2> fault on the
    //  GPU, round 5; the builtin form does not.)
    template <int K>
    __device__ __forceinline__ void gjacc4(double& a0, double& a1, double& a2, double& a3, double g) const {
        asm("s_nop 1\n\t"
            "v_fmac_f64_dpp %0, %0, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %1, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %2, %2, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %3, %3, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
            : "v"(g), "n"(K));
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
