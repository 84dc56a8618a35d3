// mpcqp_devwave.h -- the gfx950 implementation of the wave interface the kernel bodies are
// written against (lane id, wave fence, wave reductions, broadcasts).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "mpcqp_types.h"

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
    // one wavefront per problem: no team (see DevWaveT)
    static constexpr int NTEAM = 1, WV = 0;
    __device__ __forceinline__ void post(int, int = 0, int = 0, int = 0, int = 0, double = 0.0) {}
    __device__ __forceinline__ void join() {}
    // One wavefront per workgroup: LDS operations of a wave execute in issue order, so ordering
    // LDS traffic between lanes needs no s_barrier and no s_waitcnt -- only a fence the compiler
    // may not move memory operations across.
    __device__ __forceinline__ void sync() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    // The same for LDS traffic ONLY: outstanding global stores are not waited for (a kernel that streams its results
    // out and never reads them back -- K1 -- otherwise sits out a store round trip at every step of its recursion).
    __device__ __forceinline__ void sync_lds() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }

    template <int CTRL>
    static __device__ __forceinline__ double dpp(double v) {
        int lo = __double2loint(v), hi = __double2hiint(v);
        // (all rows and banks enabled and every control used here is a permutation inside a row: no lane
        // keeps its old value, so `old` is a don't-care -- passing 0 with bound_ctrl saves the copy of the
        // source into the destination that update_dpp(old = self) needs)
        lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
        hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
        return __hiloint2double(hi, lo);
    }
    static __device__ __forceinline__ double lane_value(double v, int src) {   // src wave-uniform
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
        return __hiloint2double(hi, lo);
    }
    // dpp on the rows of ROWMASK only; the other rows (and lanes without a source) read `old`
    template <int CTRL, int ROWMASK>
    static __device__ __forceinline__ double dpp_rows(double v, double old) {
        int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, ROWMASK, 0xf, false);
        int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, ROWMASK, 0xf, false);
        return __hiloint2double(hi, lo);
    }
    // `self`: x op x == x (min, max): rows outside a step's mask combine with themselves; a sum combines with 0
    template <bool SELF, class Op>
    static __device__ __forceinline__ double reduce(double v, Op op) {
        v = op(v, dpp<0xB1>(v));     // quad_perm [1,0,3,2]
        v = op(v, dpp<0x4E>(v));     // quad_perm [2,3,0,1]
        v = op(v, dpp<0x141>(v));    // row_half_mirror
        v = op(v, dpp<0x140>(v));    // row_mirror: every lane holds its 16-lane row's value
        // rows 1 and 3 take in the row before them (row_bcast:15), then rows 2 and 3 lane 31 (row_bcast:31): row 3
        // holds (r2 op r3) op (r0 op r1) -- the same pairs as a tree over four v_readlane'd row values, for two DPP
        // steps and one v_readlane pair instead of four pairs and three scalar-operand operations
        v = op(v, dpp_rows<0x142, 0xA>(v, SELF ? v : 0.0));
        v = op(v, dpp_rows<0x143, 0xC>(v, SELF ? v : 0.0));
        return lane_value(v, 63);
    }
    // sum over each aligned group of four lanes (result in all four)
    __device__ __forceinline__ double quad_sum(double v) {
        v += dpp<0xB1>(v);
        v += dpp<0x4E>(v);
        return v;
    }
    __device__ __forceinline__ double sum(double v) { return reduce<false>(v, [](double x, double y) { return x + y; }); }
    // (v_min_f64 / v_max_f64 directly: llvm.minnum / maxnum put a canonicalising v_max_f64 x, x, x in front of every operand)
    static __device__ __forceinline__ double mn_(double x, double y) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
    static __device__ __forceinline__ double mx_(double x, double y) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
    __device__ __forceinline__ double minv(double v) { return reduce<true>(v, [](double x, double y) { return mn_(x, y); }); }
    __device__ __forceinline__ double maxv(double v) { return reduce<true>(v, [](double x, double y) { return mx_(x, y); }); }
    __device__ __forceinline__ int isum(int v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    }
    __device__ __forceinline__ double bcast(double v, int src) { return lane_value(v, src); }
    // The lane id as a value the optimiser cannot trace back to threadIdx: address arithmetic built
    // on it is redone where it is used instead of being hoisted out of the interior-point loop and
    // kept in (spilled) registers for the whole kernel.
    __device__ __forceinline__ void relane() { asm volatile("" : "+v"(lane)); }
    // value of v in lane `src` (lane-varying source; any lane 0..63): two ds_bpermute_b32 through the
    // LDS crossbar, no LDS storage
    __device__ __forceinline__ double fetch(double v, int src) const {
        const int lo = __builtin_amdgcn_ds_bpermute(src << 2, __double2loint(v));
        const int hi = __builtin_amdgcn_ds_bpermute(src << 2, __double2hiint(v));
        return __hiloint2double(hi, lo);
    }
    __device__ __forceinline__ bool any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0; }
    // Substitution chain inside the 16-lane DPP rows selected by ROWS (DPP row mask, bit = row): four steps of
    //   r += (r of lane K+u of this lane's row) * c_u        u = 0..3 (REV: 3..0)
    // The broadcast is the DPP modifier of the multiply-add itself (v_fmac_f64_dpp, row_newbcast -- the one DPP control
    // 64-bit operations have); rows outside ROWS keep r.  The hardware does not interlock a DPP read of a VGPR that the
    // previous VALU instruction wrote (two wait states), and the compiler does not look inside the asm: hence the s_nop.
    template <int K, int ROWS, bool REV>
    static __device__ __forceinline__ void chain4(double& r, double c0, double c1, double c2, double c3) {
        if constexpr (!REV)
            asm("s_nop 1\n\t"
                "v_fmac_f64_dpp %0, %0, %1 row_newbcast:%5 row_mask:%9 bank_mask:0xf\n\t"
                "s_nop 1\n\t"
                "v_fmac_f64_dpp %0, %0, %2 row_newbcast:%6 row_mask:%9 bank_mask:0xf\n\t"
                "s_nop 1\n\t"
                "v_fmac_f64_dpp %0, %0, %3 row_newbcast:%7 row_mask:%9 bank_mask:0xf\n\t"
                "s_nop 1\n\t"
                "v_fmac_f64_dpp %0, %0, %4 row_newbcast:%8 row_mask:%9 bank_mask:0xf"
                : "+v"(r)
                : "v"(c0), "v"(c1), "v"(c2), "v"(c3), "n"(K), "n"(K + 1), "n"(K + 2), "n"(K + 3), "n"(ROWS));
        else
            asm("s_nop 1\n\t"
                "v_fmac_f64_dpp %0, %0, %4 row_newbcast:%8 row_mask:%9 bank_mask:0xf\n\t"
                "s_nop 1\n\t"
                "v_fmac_f64_dpp %0, %0, %3 row_newbcast:%7 row_mask:%9 bank_mask:0xf\n\t"
                "s_nop 1\n\t"
                "v_fmac_f64_dpp %0, %0, %2 row_newbcast:%6 row_mask:%9 bank_mask:0xf\n\t"
                "s_nop 1\n\t"
                "v_fmac_f64_dpp %0, %0, %1 row_newbcast:%5 row_mask:%9 bank_mask:0xf"
                : "+v"(r)
                : "v"(c0), "v"(c1), "v"(c2), "v"(c3), "n"(K), "n"(K + 1), "n"(K + 2), "n"(K + 3), "n"(ROWS));
    }
    // value of v in lane C of this lane's 16-lane row (one v_mov_b64_dpp row_newbcast)
    template <int C>
    static __device__ __forceinline__ double rowbc(double v) { return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + C, 0xf, 0xf, true); }
    // a -= (x of lane L of this lane's row) * y: the row broadcast is the multiply-add's own DPP modifier.  NOP: x may have been
    // written by the VALU instruction right before (two wait states the compiler does not see inside the asm)
    template <int L, bool NOP>
    static __device__ __forceinline__ void rowbc_fms(double& a, double x, double y) {
        if constexpr (NOP)
            asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(x), "v"(y), "n"(L));
        else
            asm("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(x), "v"(y), "n"(L));
    }
    // acc += sum_u (x of lane K+u of this lane's row) * c_u, rows of ROWS only (x is not written inside the block)
    template <int K, int ROWS>
    static __device__ __forceinline__ void fmabc4(double& acc, double x, double c0, double c1, double c2, double c3) {
        asm("s_nop 1\n\t"
            "v_fmac_f64_dpp %0, %1, %2 row_newbcast:%6 row_mask:%10 bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %0, %1, %3 row_newbcast:%7 row_mask:%10 bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %0, %1, %4 row_newbcast:%8 row_mask:%10 bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %0, %1, %5 row_newbcast:%9 row_mask:%10 bank_mask:0xf"
            : "+v"(acc)
            : "v"(x), "v"(c0), "v"(c1), "v"(c2), "v"(c3), "n"(K), "n"(K + 1), "n"(K + 2), "n"(K + 3), "n"(ROWS));
    }
};

extern __shared__ __attribute__((aligned(16))) double mpcqp_smem[];

// ---- a TEAM of T wavefronts per problem (round 6; problems beyond one row per lane, whose LDS footprint leaves SIMDs idle) ----
// Wavefront 0 runs the step exactly as the one-wavefront kernel does.  Wavefronts 1 .. T-1 are helpers: they wait at a
// workgroup barrier, read a job from the mailbox (LDS, behind the problem's carve-up), run THEIR SHARE of it -- tile rows of
// E'DE and of the panel updates of the factorisation, row slots of E v, column slots of E'w, rows of Pu'dU Pu: everything
// whose operands and results live in LDS -- and meet wavefront 0 at a second barrier.  The shares are compile-time (WV is a
// template parameter: one instantiation of the shared functions per wavefront, no run-time ownership tests); wavefront 0
// does its own share between the two barriers.  Row state, pivot chains and substitutions stay with wavefront 0.
template <int T, int WV_>
struct DevWaveT : DevWave {
    static constexpr int NTEAM = T, WV = WV_;
    double* mbox;            // 8 doubles of LDS: [0] job, args as ints in [1..2], scale in [3]
    // wavefront 0: publish a job (LDS stores of every wavefront before the barrier are visible after it)
    __device__ __forceinline__ void post(int job, int a0 = 0, int a1 = 0, int a2 = 0, int a3 = 0, double sc = 0.0) {
        if (lane == 0) {
            int* mi = reinterpret_cast<int*>(mbox);
            mi[0] = job; mi[1] = a0; mi[2] = a1; mi[3] = a2; mi[4] = a3;
            mbox[3] = sc;
        }
        __syncthreads();
    }
    __device__ __forceinline__ void join() { __syncthreads(); }
};

// specialised on compile-time dimensions
template <class SD>
// (waves_per_eu >= 2 caps the kernel at 256 registers, which makes the compiler select the VGPR
// form of v_mfma: accumulators stay in place across the K loop instead of being shuttled
// between AGPRs and VGPRs every iteration)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 8))) void k_hessian_s(Dims d, Model m) {
    DevWave w{(int)threadIdx.x};
    const SD sd(d);
    hessian_body(w, sd, m, (int)blockIdx.x, mpcqp_smem);
}

template <class SD>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MPCQP_STEP_WAVES, 8))) void k_step_s(Dims d, Model m, StepIO io) {
    DevWave w{(int)threadIdx.x};
    const SD sd(d);
    step_body(w, sd, m, io, (int)blockIdx.x, mpcqp_smem);
}

template <class DM> constexpr int auto_team();          // (mpcqp_bodies.h)

// a team of T wavefronts per problem (DevWaveT): wavefront 0 runs the step, the others serve it
template <class SD, int T, int WV>
__device__ __forceinline__ void team_run(int wv, int lane, double* mbox, const SD& sd, const Model& m, const StepIO& io) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (WV < T) {
        if (wv == WV) {
            DevWaveT<T, WV> w;
            w.lane = lane; w.mbox = mbox;
            if constexpr (WV == 0) { step_body(w, sd, m, io, (int)blockIdx.x, mpcqp_smem); w.post(TJ_EXIT); }
            else team_helper(w, sd, m, (int)blockIdx.x, mpcqp_smem);
        } else {
            team_run<SD, T, WV + 1>(wv, lane, mbox, sd, m, io);
        }
    }
#endif
}
template <class SD, int T>
__global__ __launch_bounds__(64 * T) __attribute__((amdgpu_waves_per_eu(1, 8))) void k_step_team(Dims d, Model m, StepIO io) {
    const SD sd(d);
    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    double* mbox = mpcqp_smem + make_carve(sd).total;
    team_run<SD, T, 0>(wv, (int)threadIdx.x & 63, mbox, sd, m, io);
}
#ifndef MPCQP_TEAM
#define MPCQP_TEAM 0            // wavefronts per problem of the specialised step kernel; 0: auto_team() (mpcqp_bodies.h)
#endif

inline hipError_t ensure_lds(const void* fn, size_t bytes) {
    if (bytes <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <class SD>
inline hipError_t launch_step_static(const Dims& d, const Model& m, const StepIO& io, hipStream_t st) {
    size_t lds = (size_t)make_carve(SD(d)).total * sizeof(double);
    if (const char* pad = getenv("MPCQP_LDS_PAD")) lds += (size_t)atoi(pad);   // occupancy experiments only
    constexpr int T = MPCQP_TEAM > 0 ? MPCQP_TEAM : auto_team<SD>();
    if constexpr (T > 1) {
        lds += 8 * sizeof(double);                // the team's mailbox
        hipError_t e = ensure_lds((const void*)k_step_team<SD, T>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_step_team<SD, T>), dim3(d.B), dim3(WAVE * T), lds, st, d, m, io);
        return hipGetLastError();
    } else {
        hipError_t e = ensure_lds((const void*)k_step_s<SD>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_step_s<SD>, dim3(d.B), dim3(WAVE), lds, st, d, m, io);
        return hipGetLastError();
    }
}

template <class SD>
inline hipError_t launch_hessian_static(const Dims& d, const Model& m, hipStream_t st) {
    const size_t lds = (size_t)make_carve(SD(d)).total * sizeof(double);
    hipError_t e = ensure_lds((const void*)k_hessian_s<SD>, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_hessian_s<SD>, dim3(d.B), dim3(WAVE), lds, st, d, m);
    return hipGetLastError();
}

}  // namespace mpcqp
