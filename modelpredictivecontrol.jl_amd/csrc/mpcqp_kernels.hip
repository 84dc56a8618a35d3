// mpcqp_kernels.hip -- gfx950 kernels of the batched LinMPC step: one QP per 64-lane wavefront,
// one wavefront per workgroup, the condensed problem (step-response table, packed normal-equation
// tile, residual/row state) staged in LDS.  Bodies: mpcqp_bodies.h.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <map>
#include <mutex>
#include <string>
#include <tuple>

#include "mpcqp_bodies.h"
#include "mpcqp_devwave.h"
#include "mpcqp_dispatch.h"
#include "mpcqp_launch.h"

namespace mpcqp {

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
hipError_t launch_predmat(const Dims& d, const Model& m, bool terminal, hipStream_t st) {
    size_t lds = (size_t)predmat_lds_doubles(d) * sizeof(double);
    hipError_t e = ensure_lds((const void*)k_predmat, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_predmat, dim3(d.B), dim3(WAVE), lds, st, d, m, terminal ? 1 : 0);
    return hipGetLastError();
}

static bool env_flag(const char* name, bool dflt) {
    const char* e = getenv(name);
    return e ? e[0] == '1' : dflt;
}
static bool force_generic() { static const bool f = env_flag("MPCQP_FORCE_GENERIC", false); return f; }

// ---- on-demand specialisation ----------------------------------------------------------------
// Dimensions outside the ahead-of-time list get their own compile-time-dims kernel the first
// time they are used: csrc/mpcqp_spec.hip is compiled with the installation's hipcc (the same
// compiler as the ahead-of-time build), cached as lib/spec_cache/spec_<dims>.so and dlopen'ed.
// MPCQP_JIT=0 disables it (generic runtime-dims kernel then); any failure falls back to the
// generic kernel with one message on stderr.
struct SpecLib {
    int (*matches)(const Dims*) = nullptr;
    int (*matches_dims)(const Dims*) = nullptr;
    int (*step)(const Dims*, const Model*, const StepIO*, void*) = nullptr;
    int (*hessian)(const Dims*, const Model*, void*) = nullptr;
};
using SpecKey = std::tuple<int, int, int, int, int, int, unsigned, int>;
static std::mutex g_spec_mu;
static std::map<SpecKey, SpecLib> g_spec;        // failed builds are cached as empty entries

static std::string lib_dir() {
    Dl_info info;
    if (!dladdr((const void*)&lib_dir, &info) || !info.dli_fname) return ".";
    std::string p(info.dli_fname);
    size_t s = p.rfind('/');
    return s == std::string::npos ? "." : p.substr(0, s);
}

static const SpecLib* jit_specialise(const Dims& d) {
    static const bool enabled = env_flag("MPCQP_JIT", true);
    if (!enabled) return nullptr;
    const SpecKey key{d.nu, d.ny, d.nxh, d.Hp, d.Hc, d.neps, d.gmask, d.default_nb};
    std::lock_guard<std::mutex> lock(g_spec_mu);
    auto it = g_spec.find(key);
    if (it != g_spec.end()) return it->second.step ? &it->second : nullptr;
    SpecLib sl;
    const std::string dir = lib_dir(), cache = dir + "/spec_cache", src = dir + "/../csrc";
    char name[160];
    snprintf(name, sizeof name, "spec_r%d_%d_%d_%d_%d_%d_%d_%x_%d.so", MPCQP_KERNEL_REV, d.nu, d.ny, d.nxh, d.Hp, d.Hc, d.neps, d.gmask, d.default_nb);
    const std::string so = cache + "/" + name;
    struct stat sb;
    if (stat(so.c_str(), &sb) != 0) {
        mkdir(cache.c_str(), 0755);
        const char* hipcc = getenv("HIPCC") ? getenv("HIPCC") : "/opt/rocm/bin/hipcc";
        char cmd[2048];
        snprintf(cmd, sizeof cmd,
                 "%s --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -w -I%s "
                 "-DMPCQP_SPEC_DIMS=%d,%d,%d,%d,%d,%d,%uu,%d %s/mpcqp_spec.hip -o %s.tmp%d 2>&1 && mv %s.tmp%d %s",
                 hipcc, src.c_str(), d.nu, d.ny, d.nxh, d.Hp, d.Hc, d.neps, d.gmask, d.default_nb, src.c_str(),
                 so.c_str(), (int)getpid(), so.c_str(), (int)getpid(), so.c_str());   // (ranks of one job may build the same object)
        fprintf(stderr, "[mpcqp] specialising the step kernel for nu=%d ny=%d nxhat=%d Hp=%d Hc=%d "
                        "neps=%d rows=0x%x (one-time, cached in %s)\n",
                d.nu, d.ny, d.nxh, d.Hp, d.Hc, d.neps, d.gmask, cache.c_str());
        FILE* p = popen(cmd, "r");
        std::string out;
        if (p) {
            char buf[512];
            while (fgets(buf, sizeof buf, p)) out += buf;
            const int rc = pclose(p);
            if (rc != 0) fprintf(stderr, "[mpcqp] specialisation failed (rc=%d), using the generic kernel:\n%s\n", rc, out.c_str());
        }
    }
    if (stat(so.c_str(), &sb) == 0) {
        void* hdl = dlopen(so.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (hdl) {
            sl.matches = (int (*)(const Dims*))dlsym(hdl, "mpcqp_spec_matches");
            sl.matches_dims = (int (*)(const Dims*))dlsym(hdl, "mpcqp_spec_matches_dims");
            sl.step = (int (*)(const Dims*, const Model*, const StepIO*, void*))dlsym(hdl, "mpcqp_spec_launch_step");
            sl.hessian = (int (*)(const Dims*, const Model*, void*))dlsym(hdl, "mpcqp_spec_launch_hessian");
            if (!sl.matches || !sl.step || !sl.hessian || !sl.matches(&d)) sl = SpecLib{};
        } else {
            fprintf(stderr, "[mpcqp] dlopen(%s) failed: %s\n", so.c_str(), dlerror());
        }
    }
    auto res = g_spec.emplace(key, sl);
    return res.first->second.step ? &res.first->second : nullptr;
}

hipError_t launch_hessian(const Dims& d, const Model& m, hipStream_t st) {
    // (a block-diagonal M_Hp takes the scalar contraction of the runtime-dims kernel: set-up path)
    if (!force_generic() && !m.Mblk) {
#define X(NU, NY, NXH, HP, HC, NEPS, GM)                                            \
        {                                                                           \
            using SD = StaticDims<NU, NY, NXH, HP, HC, NEPS, GM>;                   \
            if (SD::matches_dims(d)) return launch_hessian_static<SD>(d, m, st);    \
        }
        MPCQP_SPECIALIZATIONS(X)
#undef X
    }
    // (the Hessian of other dimensions stays on the generic kernel: it runs once per set_model)
    size_t lds = (size_t)make_carve(d).total * sizeof(double);
    hipError_t e = ensure_lds((const void*)k_hessian, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_hessian, dim3(d.B), dim3(WAVE), lds, st, d, m);
    return hipGetLastError();
}

hipError_t launch_step(const Dims& d, const Model& m, const StepIO& io, hipStream_t st) {
    if (!force_generic()) {
#define X(NU, NY, NXH, HP, HC, NEPS, GM)                                            \
        {                                                                           \
            using SD = StaticDims<NU, NY, NXH, HP, HC, NEPS, GM>;                   \
            if (SD::matches(d)) return launch_step_static<SD>(d, m, io, st);        \
        }
        MPCQP_SPECIALIZATIONS(X)
#undef X
        if (d.nw == 0 && d.nZ <= WAVE)   // (custom linear constraints and nZ~ > 64 run on the runtime-dims kernel)
            if (const SpecLib* sl = jit_specialise(d)) return (hipError_t)sl->step(&d, &m, &io, (void*)st);
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
