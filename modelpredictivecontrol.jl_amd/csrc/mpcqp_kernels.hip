// mpcqp_kernels.hip -- gfx950 kernels of the batched LinMPC step: one QP per 64-lane wavefront,
// one wavefront per workgroup, the condensed problem (step-response table, packed normal-equation
// tile, residual/row state) staged in LDS.  Bodies: mpcqp_bodies.h.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include <dirent.h>
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "mpcqp_bodies.h"
#include "mpcqp_devwave.h"
#include "mpcqp_dispatch.h"
#include "mpcqp_launch.h"

extern char** environ;      // (the specialisation compiler inherits the environment)

namespace mpcqp {

#ifndef MPCQP_K1_MFMA
#define MPCQP_K1_MFMA 1           // 0: the LDS loops for every shape (timing experiments)
#endif
#ifndef MPCQP_K1_MFMA_MIN_NX
#define MPCQP_K1_MFMA_MIN_NX 10   // below, the 16-wide tiles are mostly padding and the LDS loops are faster (C2, nx̂ = 6: 0.55 vs 0.66 ms)
#endif
static bool predmat_on_mfma(const Dims& d) {
    static const int min_nx = [] {          // (MPCQP_K1_MFMA_MIN_NX=1 in the environment: every eligible shape, for the tests)
        const char* e = getenv("MPCQP_K1_MFMA_MIN_NX");
        return e && atoi(e) > 0 ? atoi(e) : MPCQP_K1_MFMA_MIN_NX;
    }();
    return MPCQP_K1_MFMA && predmat_mfma_ok(d) && d.nxh >= min_nx;
}
__global__ __launch_bounds__(64) void k_predmat(Dims d, Model m, int terminal) {
    DevWave w{(int)threadIdx.x};
    predmat_body(w, d, m, (int)blockIdx.x, mpcqp_smem, terminal != 0);
}
// the same tables from chains of v_mfma_f64_16x16x4 (predmat_mfma: no LDS, the state stays in the accumulators)
__global__ __launch_bounds__(64) void k_predmat_mfma(Dims d, Model m, int terminal) {
    predmat_mfma((int)threadIdx.x, d, m, (int)blockIdx.x, terminal != 0);
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
    if (predmat_on_mfma(d)) {
        hipLaunchKernelGGL(k_predmat_mfma, dim3(d.B), dim3(WAVE), 0, st, d, m, terminal ? 1 : 0);
        return hipGetLastError();
    }
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
// Dimensions outside the ahead-of-time list get their own compile-time-dims kernel: csrc/mpcqp_spec.hip
// is compiled with the installation's hipcc (the same compiler as the ahead-of-time build), cached as
// spec_<rev>_<dims>.so and dlopen'ed.  The compilation NEVER happens inside a step: it is done by
// prepare_step() (mpcqp_prepare / mpcqp_prebuild of the C-ABI), the step only looks the object up and
// runs the runtime-dimension kernel when there is none.  The compiler is started with posix_spawn on an
// argument vector (no shell: paths with spaces or metacharacters are data), its output goes to a log
// file next to the object.
//   MPCQP_JIT=0          no on-demand kernels at all
//   HIPCC                the compiler binary (one path; default /opt/rocm/bin/hipcc)
//   MPCQP_JIT_FLAGS      extra compiler arguments, space separated (experiments)
//   MPCQP_CACHE_DIR      where the objects live; default <library dir>/spec_cache when that is writable,
//                        else $XDG_CACHE_HOME/mpcqp or ~/.cache/mpcqp
struct SpecLib {
    int (*matches)(const Dims*) = nullptr;
    int (*matches_dims)(const Dims*) = nullptr;
    int (*step)(const Dims*, const Model*, const StepIO*, void*) = nullptr;
    int (*hessian)(const Dims*, const Model*, void*) = nullptr;
    bool verified = false;        // compared with the runtime-dimension kernel on this machine (marker <object>.ok)
    bool no_marker = false;       // the marker was looked for and not found: steps do not stat() again (mpcqp_prepare does)
    std::string path;             // the object this entry was loaded from
};
using SpecKey = std::tuple<int, int, int, int, int, int, unsigned, int>;
// last field (also the last field of the object name): bit 0 default move blocking, bit 1 dense M_Hp / L_Hp in the gradient,
// bits 2.. the number of custom linear constraint rows per step
static int spec_variant(const Dims& d) { return d.default_nb | (d.dense_w ? 2 : 0) | (d.nw << 2); }
static SpecKey spec_key(const Dims& d) { return SpecKey{d.nu, d.ny, d.nxh, d.Hp, d.Hc, d.neps, d.gmask, spec_variant(d)}; }
static std::mutex g_spec_mu;
static std::map<SpecKey, SpecLib> g_spec;        // failed loads are cached as empty entries
static const SpecLib* find_verified_spec(const Dims& d);

static std::string lib_dir() {
    Dl_info info;
    if (!dladdr((const void*)&lib_dir, &info) || !info.dli_fname) return ".";
    std::string p(info.dli_fname);
    size_t s = p.rfind('/');
    return s == std::string::npos ? "." : p.substr(0, s);
}

static bool dir_writable(const std::string& d) {
    struct stat sb;
    if (stat(d.c_str(), &sb) != 0) {
        if (mkdir(d.c_str(), 0755) != 0 && errno != EEXIST) return false;
    }
    return access(d.c_str(), W_OK | X_OK) == 0;
}

static std::string user_cache_dir() {
    std::string base;
    if (const char* x = getenv("XDG_CACHE_HOME")) base = x;
    if (base.empty()) {
        const char* home = getenv("HOME");
        if (!home || !home[0]) return "";        // no private place to keep shared objects
        base = std::string(home) + "/.cache";
        (void)dir_writable(base);
    }
    const std::string d = base + "/mpcqp";
    (void)dir_writable(d);
    return d;
}

// where NEW files go (objects built here, the .ok / .rejected markers of objects in a read-only directory)
static std::string cache_dir() {
    if (const char* e = getenv("MPCQP_CACHE_DIR")) {
        if (e[0]) { (void)dir_writable(e); return e; }
    }
    const std::string local = lib_dir() + "/spec_cache";
    if (dir_writable(local)) return local;
    return user_cache_dir();
}

// where objects are LOOKED UP: the installation's own <library dir>/spec_cache is searched even when it is read-only
// (the normal deployment: objects shipped by `python -m mpcqp.prebuild` / mpcqp_prebuild on a build host), then the
// user's cache.  MPCQP_CACHE_DIR replaces both.
static std::vector<std::string> search_dirs() {
    std::vector<std::string> v;
    if (const char* e = getenv("MPCQP_CACHE_DIR")) {
        if (e[0]) { v.push_back(e); return v; }
    }
    v.push_back(lib_dir() + "/spec_cache");
    const std::string u = user_cache_dir();
    if (!u.empty() && u != v[0]) v.push_back(u);
    return v;
}

// Shared objects are only loaded from a directory that belongs to this user (or root) and that others cannot write.
static bool cache_dir_trusted(const std::string& d) {
    struct stat sb;
    if (d.empty() || stat(d.c_str(), &sb) != 0 || !S_ISDIR(sb.st_mode)) return false;
    if (sb.st_uid != geteuid() && sb.st_uid != 0) return false;
    const bool ok = (sb.st_mode & (S_IWGRP | S_IWOTH)) == 0;
    if (!ok) {
        static bool told = false;
        if (!told) {
            told = true;
            fprintf(stderr, "[mpcqp] specialisation cache %s is writable by group / others: not used (chmod go-w, or set MPCQP_CACHE_DIR)\n", d.c_str());
        }
    }
    return ok;
}

// ... and only files that belong to this user (or root) and that others cannot write
static bool object_trusted(const std::string& so) {
    struct stat sb;
    if (stat(so.c_str(), &sb) != 0 || !S_ISREG(sb.st_mode)) return false;
    if (sb.st_uid != geteuid() && sb.st_uid != 0) return false;
    return (sb.st_mode & (S_IWGRP | S_IWOTH)) == 0;
}

static std::string hipcc_path() {
    const char* hipcc = getenv("HIPCC");
    return hipcc && hipcc[0] ? hipcc : "/opt/rocm/bin/hipcc";
}

// identity of the compiler the on-demand kernels are built with (size and mtime of the binary): part of the
// object name, so that a cache filled by another hipcc is not reused WHEN THIS MACHINE CAN BUILD ITS OWN (the same
// 32-bit value as __graft_entry__.py computes for the objects it pre-builds).  0: no compiler here.
static unsigned compiler_id() {
    struct stat sb;
    if (stat(hipcc_path().c_str(), &sb) != 0) return 0u;
    return (unsigned)(((unsigned long long)sb.st_size * 1000003ull) ^ (unsigned long long)sb.st_mtime);
}

static std::string spec_dims_suffix(const Dims& d) {
    char name[160];
    snprintf(name, sizeof name, "_%d_%d_%d_%d_%d_%d_%x_%d.so", d.nu, d.ny, d.nxh, d.Hp, d.Hc, d.neps, d.gmask, spec_variant(d));
    return name;
}
static std::string spec_rev_prefix() { return "spec_r" + std::to_string(MPCQP_KERNEL_REV) + "_c"; }

// name of the object this machine's compiler builds
static std::string spec_name(const Dims& d) {
    char cid[16];
    snprintf(cid, sizeof cid, "%08x", compiler_id());
    return spec_rev_prefix() + cid + spec_dims_suffix(d);
}

// objects rejected by this process whose file could not be renamed (read-only cache)
static std::mutex g_rej_mu;
static std::vector<std::string> g_rejected;
static bool rejected_here(const std::string& so) {
    std::lock_guard<std::mutex> lock(g_rej_mu);
    for (const std::string& r : g_rejected) if (r == so) return true;
    struct stat sb;
    const size_t sl = so.rfind('/');
    return stat((cache_dir() + "/" + so.substr(sl + 1) + ".rejected").c_str(), &sb) == 0;
}

// The object of `d` a step may load: the one this machine's compiler builds (exact name) in any search directory, else
// -- no compiler here, or a cache filled by a build host with another hipcc -- any object of the same kernel revision and
// dimensions, `spec_r<rev>_c*_<dims>.so`.  Safe because no object runs before mpcqp_prepare has compared it with the
// runtime-dimension kernel ON THIS MACHINE (marker .ok); with a local compiler the exact name wins and a foreign object
// is only used while the local one does not exist (prepare builds it).  "" if there is none.
static std::string locate_spec(const Dims& d, bool foreign_ok) {
    const std::string exact = spec_name(d), pre = spec_rev_prefix(), suf = spec_dims_suffix(d);
    const std::vector<std::string> dirs = search_dirs();
    for (const std::string& dir : dirs) {
        if (!cache_dir_trusted(dir)) continue;
        const std::string so = dir + "/" + exact;
        if (object_trusted(so) && !rejected_here(so)) return so;
    }
    if (!foreign_ok) return "";
    for (const std::string& dir : dirs) {
        if (!cache_dir_trusted(dir)) continue;
        DIR* dp = opendir(dir.c_str());
        if (!dp) continue;
        std::string best;
        while (struct dirent* e = readdir(dp)) {
            const std::string f = e->d_name;
            if (f.size() != exact.size() || f.compare(0, pre.size(), pre) != 0 ||
                f.compare(f.size() - suf.size(), suf.size(), suf) != 0)
                continue;
            const std::string so = dir + "/" + f;
            if (object_trusted(so) && !rejected_here(so) && (best.empty() || so < best)) best = so;
        }
        closedir(dp);
        if (!best.empty()) return best;
    }
    return "";
}
static std::string locate_spec(const Dims& d) { return locate_spec(d, compiler_id() == 0u || !env_flag("MPCQP_CACHE_STRICT", false)); }

// marker <object>.ok next to the object when its directory is writable, else under the same name in cache_dir()
static std::string marker_path(const std::string& so, const char* ext) {
    const size_t sl = so.rfind('/');
    const std::string dir = so.substr(0, sl);
    if (access(dir.c_str(), W_OK | X_OK) == 0) return so + ext;
    return cache_dir() + "/" + so.substr(sl + 1) + ext;
}

static bool jit_enabled() {
    static const bool on = env_flag("MPCQP_JIT", true);
    return on;
}

// (custom linear constraints and dense M_Hp / L_Hp get on-demand variants of their own: -DMPCQP_SPEC_NW=nw compiles the
// custom rows in (row state in registers like every other group), -DMPCQP_SPEC_DENSE the dense products of the gradient;
// compile-time dims up to MPCQP_SPEC_NZMAX:
// beyond one row per lane the specialisation keeps the several-rows-per-lane factorisation of the runtime dims but has
// the matrix-core E'DE, register rows and constant trip counts)
#ifndef MPCQP_SPEC_NZMAX
#define MPCQP_SPEC_NZMAX 256      // (round 5: 128 before.  Whatever fits the LDS -- nZ~ up to ~185 -- gets its specialisation: nZ~ = 141 ... 161 run 3.6 - 4.3
                                  //  times faster than on the runtime-dimension kernel, 450 - 470 VGPRs at one wavefront per SIMD, compiled in ~50 s)
#endif
static bool spec_eligible(const Dims& d) {
    return d.nZ <= MPCQP_SPEC_NZMAX;
}

// run `argv` (argv[0] = binary), stdout+stderr appended to `log`; returns the exit status, -1 on failure to start
static int run_process(const std::vector<std::string>& argv, const std::string& log) {
    std::vector<char*> av;
    for (const std::string& a : argv) av.push_back(const_cast<char*>(a.c_str()));
    av.push_back(nullptr);
    posix_spawn_file_actions_t fa;
    posix_spawn_file_actions_init(&fa);
    posix_spawn_file_actions_addopen(&fa, 1, log.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    posix_spawn_file_actions_adddup2(&fa, 1, 2);
    pid_t pid = 0;
    const int rc = posix_spawn(&pid, av[0], &fa, nullptr, av.data(), ::environ);
    posix_spawn_file_actions_destroy(&fa);
    if (rc != 0) return -1;
    int status = 0;
    while (waitpid(pid, &status, 0) < 0) {
        if (errno != EINTR) return -1;
    }
    return WIFEXITED(status) ? WEXITSTATUS(status) : -1;
}

// compile the specialisation of `d` into the cache unless it is there already; 0 = present afterwards
static int build_spec(const Dims& d, std::string* path_out, std::string* err) {
    const std::string cache = cache_dir(), so = cache + "/" + spec_name(d);
    if (path_out) *path_out = so;
    if (!cache_dir_trusted(cache)) {
        if (err) *err = "specialisation cache directory '" + cache + "' is missing, not owned by this user or writable by others (set MPCQP_CACHE_DIR)";
        return -1;
    }
    {
        // an object of this shape that may be loaded exists already (this machine's own, or -- without a compiler here --
        // one a build host shipped)
        const std::string have = locate_spec(d, compiler_id() == 0u);
        if (!have.empty()) { if (path_out) *path_out = have; return 0; }
    }
    if (compiler_id() == 0u) {
        if (err) *err = "no compiler (" + hipcc_path() + ") and no prebuilt object of this shape in the specialisation cache";
        return -1;
    }
    if (access(cache.c_str(), W_OK | X_OK) != 0) {
        if (err) *err = "specialisation cache directory " + cache + " is not writable (set MPCQP_CACHE_DIR)";
        return -1;
    }
    const std::string src = lib_dir() + "/../csrc";
    char dims[160];
    snprintf(dims, sizeof dims, "-DMPCQP_SPEC_DIMS=%d,%d,%d,%d,%d,%d,%uu,%d", d.nu, d.ny, d.nxh, d.Hp, d.Hc, d.neps,
             d.gmask, d.default_nb);
    const std::string tmp = so + ".tmp" + std::to_string((long)getpid());   // (ranks of one job may build the same object)
    std::vector<std::string> argv = {hipcc_path(), "--offload-arch=gfx950", "-O3",
                                     "-std=c++17", "-shared", "-fPIC", "-w", "-I" + src, dims};
    // beyond one row per lane the LDS footprint (Phi alone is 47 KB at nZ~ = 106) leaves at most one wavefront per SIMD:
    // the kernel may as well use the whole register file (row state of several rows per lane in registers, no spills)
    if (d.nZ > WAVE) argv.push_back("-DMPCQP_STEP_WAVES=1");
    // `#pragma unroll` gives up beyond 16k unrolled instructions (LLVM's pragma-unroll-threshold): from eight tile rows on
    // (nZ~ >= ~113) a loop over the tile rows stayed rolled, its accumulator / pointer arrays went to scratch memory and every
    // LDS and global access of the kernel became a flat one -- found in round 5 from the counters of the nZ~ = 151 kernel
    // (23 LDS instructions per solve).  With the threshold lifted: nZ~ = 125 24.3 -> 15.5 ms, 151 37.3 -> 23.7 ms, 161 39.4 ->
    // 25.8 ms per 2048 controllers (no scratch, no flat access); smaller shapes compile to the same code.
    argv.push_back("-mllvm");
    argv.push_back("-pragma-unroll-threshold=1048576");
    if (d.dense_w) argv.push_back("-DMPCQP_SPEC_DENSE=1");
    if (d.nw > 0) argv.push_back("-DMPCQP_SPEC_NW=" + std::to_string(d.nw));
    if (const char* extra = getenv("MPCQP_JIT_FLAGS")) {
        std::string tok;
        for (const char* c = extra;; ++c) {
            if (*c == ' ' || *c == 0) { if (!tok.empty()) argv.push_back(tok); tok.clear(); if (!*c) break; }
            else tok += *c;
        }
    }
    argv.push_back(src + "/mpcqp_spec.hip");
    argv.push_back("-o");
    argv.push_back(tmp);
    fprintf(stderr, "[mpcqp] specialising the step kernel for nu=%d ny=%d nxhat=%d Hp=%d Hc=%d neps=%d rows=0x%x "
                    "(one-time, cached in %s)\n", d.nu, d.ny, d.nxh, d.Hp, d.Hc, d.neps, d.gmask, cache.c_str());
    const std::string log = so + ".log";
    const int rc = run_process(argv, log);
    if (rc != 0 || rename(tmp.c_str(), so.c_str()) != 0) {
        (void)unlink(tmp.c_str());
        if (err) *err = "compiling the specialisation failed (exit " + std::to_string(rc) + ", see " + log + ")";
        return -1;
    }
    (void)unlink(log.c_str());
    return 0;
}

// the loaded specialisation of `d`, or nullptr; loads a cached object, never compiles
static const SpecLib* find_spec(const Dims& d, bool load) {
    if (!jit_enabled() || !spec_eligible(d)) return nullptr;
    const SpecKey key = spec_key(d);
    std::lock_guard<std::mutex> lock(g_spec_mu);
    auto it = g_spec.find(key);
    if (it != g_spec.end()) return it->second.step ? &it->second : nullptr;
    if (!load) return nullptr;
    const std::string so = locate_spec(d);
    if (so.empty()) {
        // not built (yet).  The miss is remembered (a step of a small dense-row handle asked again at every launch --
        // stat / opendir / readdir over the cache directories under g_spec_mu, and a dlopen inside a step once another
        // process had filled the cache); mpcqp_prepare forgets it before it builds or looks again (prepare_step_other).
        g_spec.emplace(key, SpecLib{});
        return nullptr;
    }
    SpecLib sl;
    void* hdl = dlopen(so.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (hdl) {
        sl.matches = (int (*)(const Dims*))dlsym(hdl, "mpcqp_spec_matches");
        sl.matches_dims = (int (*)(const Dims*))dlsym(hdl, "mpcqp_spec_matches_dims");
        sl.step = (int (*)(const Dims*, const Model*, const StepIO*, void*))dlsym(hdl, "mpcqp_spec_launch_step");
        sl.hessian = (int (*)(const Dims*, const Model*, void*))dlsym(hdl, "mpcqp_spec_launch_hessian");
        if (!sl.matches || !sl.step || !sl.hessian || !sl.matches(&d)) sl = SpecLib{};
        else sl.path = so;
    } else {
        fprintf(stderr, "[mpcqp] dlopen(%s) failed: %s\n", so.c_str(), dlerror());
    }
    auto res = g_spec.emplace(key, sl);
    return res.first->second.step ? &res.first->second : nullptr;
}

static bool aot_matches(const Dims& d) {
#define X(NU, NY, NXH, HP, HC, NEPS, GM) \
    if (StaticDims<NU, NY, NXH, HP, HC, NEPS, GM>::matches(d)) return true;
    MPCQP_SPECIALIZATIONS(X)
#undef X
    return false;
}

// Which kernel a step of `d` runs on: 0 the runtime-dimension kernel, 1 an ahead-of-time
// specialisation, 2 an on-demand one (already loaded or loadable from the cache).
// The small-problem kernel takes the step -- unless it is its dense-row variant on a grid of the latency regime (at most one
// controller per SIMD as one-per-wavefront grid: B <= 1024) and the handle's own specialisation is there: measured on C2
// dimensions with a soft ymax (40 dense rows; scripts/small_vs_wave.py 4,2,2,20,5), one controller per wavefront with the
// matrix-core assembly and the polish is 1.5 times faster at B <= 1024 (0.25 against 0.38 ms), the two tie from 4096 on.
// MPCQP_SMALL_Y=1 keeps the dense-row variant whatever the batch (tests of that kernel).
static bool small_takes(const Dims& d, const Model& m, const StepIO& io) {
    if (force_generic() || !small_eligible(d, m, io)) return false;
    const char* e = getenv("MPCQP_SMALL_Y");           // (read at every call: tests switch it inside one process)
    const bool keep_y = e && e[0] == '1';
    if (small_has_y(d) && !keep_y && d.B <= 1024 && ((!d.dense_w && aot_matches(d)) || find_verified_spec(d))) return false;
    return true;
}

int step_kernel_kind(const Dims& d, const Model& m) {
    if (small_takes(d, m, StepIO{})) return 3;
    return step_kernel_kind_other(d);
}

// the kernel of the steps the small-problem kernel does not take (Ŷ requested, fused Kalman steps)
int step_kernel_kind_other(const Dims& d) {
    if (force_generic()) return 0;
    if (!d.dense_w && aot_matches(d)) return 1;     // (the kernels compiled into the library have no dense-weight products)
    return find_verified_spec(d) ? 2 : 0;
}

// Make the specialised kernel of `d` available (compile if needed, load).  Returns the kernel kind
// as step_kernel_kind(); `err` receives the reason when an eligible specialisation could not be built.
static int prepare_step_other(const Dims& d, std::string* err) {
    if (force_generic()) return 0;
    if (!d.dense_w && aot_matches(d)) return 1;
    if (!jit_enabled() || !spec_eligible(d)) return 0;
    {
        std::lock_guard<std::mutex> lock(g_spec_mu);            // a remembered miss (find_spec) is looked up again by a prepare
        auto it = g_spec.find(spec_key(d));
        if (it != g_spec.end() && !it->second.step) g_spec.erase(it);
    }
    if (find_spec(d, true)) return 2;
    if (build_spec(d, nullptr, err) != 0) return 0;
    {
        std::lock_guard<std::mutex> lock(g_spec_mu);            // forget a remembered failure of an earlier load
        g_spec.erase(spec_key(d));
    }
    if (find_spec(d, true)) return 2;
    if (err) *err = "the specialisation was built but could not be loaded";
    return 0;
}

int prepare_step(const Dims& d, const Model& m, std::string* err) {
    // (the kernel behind the small one is prepared as well: steps that ask for Ŷ or fuse the Kalman steps run on it)
    const int other = prepare_step_other(d, err);
    return small_takes(d, m, StepIO{}) ? 3 : other;
}

// compile only (no device, no load): for build pipelines
int prebuild_step(const Dims& d, std::string* err) {
    if (aot_matches(d)) return 1;
    if (!jit_enabled() || !spec_eligible(d)) return 0;
    return build_spec(d, nullptr, err) == 0 ? 2 : -1;
}

// A freshly built specialisation is checked once against the runtime-dimension kernel (mpcqp_prepare, host side)
// before it is trusted: `<object>.ok` records that it passed, a failing object is renamed `<object>.bad` and never
// loaded again (the local hipcc builds these kernels: a compiler that miscompiles them must not go unnoticed).
bool spec_present(const Dims& d) { return !force_generic() && jit_enabled() && spec_eligible(d) && find_spec(d, true) != nullptr; }

bool spec_verified(const Dims& d) {
    const SpecLib* sl = find_spec(d, true);
    if (!sl) return false;
    if (sl->verified) return true;
    struct stat sb;
    const bool ok = stat((sl->path + ".ok").c_str(), &sb) == 0 || stat(marker_path(sl->path, ".ok").c_str(), &sb) == 0;
    std::lock_guard<std::mutex> lock(g_spec_mu);
    auto it = g_spec.find(spec_key(d));
    if (it != g_spec.end() && it->second.step) { it->second.verified = ok; it->second.no_marker = !ok; }
    return ok;
}
void mark_spec_verified(const Dims& d) {
    const SpecLib* sl = find_spec(d, true);
    if (!sl) return;
    // (a marker that cannot be written -- read-only cache and no user cache -- only costs the comparison again in the
    //  next process: this process remembers)
    if (FILE* f = fopen(marker_path(sl->path, ".ok").c_str(), "w")) fclose(f);
    std::lock_guard<std::mutex> lock(g_spec_mu);
    auto it = g_spec.find(spec_key(d));
    if (it != g_spec.end() && it->second.step) { it->second.verified = true; it->second.no_marker = false; }
}
// the specialisation a STEP may run: loaded AND verified (mpcqp_prepare's self-test, or the marker of an earlier one);
// an object that some other process, a build pipeline (mpcqp_prebuild) or a prepare without a model put into the cache
// runs only after this machine has compared it with the runtime-dimension kernel.  The marker is looked for once per
// loaded object (and again by every mpcqp_prepare): a step does no file-system work.
static const SpecLib* find_verified_spec(const Dims& d) {
    const SpecLib* sl = find_spec(d, true);
    if (!sl) return nullptr;
    if (sl->verified) return sl;
    if (sl->no_marker) return nullptr;
    return spec_verified(d) ? sl : nullptr;
}

void reject_spec(const Dims& d) {
    std::string so;
    if (const SpecLib* sl = find_spec(d, true)) so = sl->path;
    if (!so.empty() && rename(so.c_str(), (so + ".bad").c_str()) != 0) {
        // read-only cache: remember the rejection in this process and, when there is a writable cache, for the next ones
        { std::lock_guard<std::mutex> lock(g_rej_mu); g_rejected.push_back(so); }
        if (FILE* f = fopen(marker_path(so, ".rejected").c_str(), "w")) fclose(f);
    }
    std::lock_guard<std::mutex> lock(g_spec_mu);
    g_spec[spec_key(d)] = SpecLib{};
}

hipError_t launch_step_generic(const Dims& d, const Model& m, const StepIO& io, hipStream_t st) {
    size_t lds = (size_t)make_carve(d).total * sizeof(double);
    hipError_t e = ensure_lds((const void*)k_step, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_step, dim3(d.B), dim3(WAVE), lds, st, d, m, io);
    return hipGetLastError();
}

hipError_t launch_hessian(const Dims& d, const Model& m, hipStream_t st) {
    // (a block-diagonal M_Hp takes the scalar contraction of the runtime-dims kernel: set-up path)
    if (!force_generic() && !m.Mblk && !m.Mfull && !m.Ndense && !m.Ldense) {
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
    if (small_takes(d, m, io)) return launch_step_small(d, m, io, st);
    return launch_step_spec_or_aot(d, m, io, st);
}

// the one-QP-per-wavefront kernels: ahead-of-time specialisation, on-demand specialisation, runtime dimensions
hipError_t launch_step_spec_or_aot(const Dims& d, const Model& m, const StepIO& io, hipStream_t st) {
    if (!force_generic()) {
        if (!d.dense_w) {      // (dense M_Hp / L_Hp: on-demand variant or the runtime-dimension kernel)
#define X(NU, NY, NXH, HP, HC, NEPS, GM)                                            \
        {                                                                           \
            using SD = StaticDims<NU, NY, NXH, HP, HC, NEPS, GM>;                   \
            if (SD::matches(d)) return launch_step_static<SD>(d, m, io, st);        \
        }
        MPCQP_SPECIALIZATIONS(X)
#undef X
        }
        if (const SpecLib* sl = find_verified_spec(d)) return (hipError_t)sl->step(&d, &m, &io, (void*)st);
    }
    size_t lds = (size_t)make_carve(d).total * sizeof(double);
    hipError_t e = ensure_lds((const void*)k_step, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_step, dim3(d.B), dim3(WAVE), lds, st, d, m, io);
    return hipGetLastError();
}

// the on-demand specialisation itself, verified or not: only mpcqp_prepare's comparison calls this
hipError_t launch_step_unverified_spec(const Dims& d, const Model& m, const StepIO& io, hipStream_t st) {
    if (const SpecLib* sl = find_spec(d, true)) return (hipError_t)sl->step(&d, &m, &io, (void*)st);
    return launch_step_spec_or_aot(d, m, io, st);
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
