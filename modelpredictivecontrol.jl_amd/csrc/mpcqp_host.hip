// mpcqp_host.hip -- C-ABI of include/mpcqp.h: handle, device residency, launches.
// There is deliberately NO CPU fallback in this library: every compute entry point needs a HIP
// device and fails with MPCQP_ERR_DEVICE otherwise.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mpcqp.h"
#include "mpcqp_hostutil.h"
#include "mpcqp_launch.h"
#include "ms_bodies.h"
#include "ms_launch.h"

using namespace mpcqp;

thread_local std::string g_hip_err;

struct DBuf {                      // owned device array of doubles (or ints)
    void* p = nullptr;
    size_t bytes = 0;
};

struct mpcqp_handle_s {
    Dims d{};
    Model m{};
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_s0 = nullptr, ev_s1 = nullptr, ev_c0 = nullptr, ev_c1 = nullptr, ev_cm = nullptr;
    bool step_timed = false, cond_timed = false;
    bool have_model = false, have_weights = false, terminal_built = false;
    // Model::Hpk holds K2's result for the CURRENT tables and weights.  Cleared whenever an input of K2 changes or its
    // launch is skipped (the problem did not fit the LDS at that moment: condensed_fits depends on the row groups, which
    // set_bounds / set_custom_bounds may shrink later); ensure_hessian() launches K2 lazily wherever H~ is about to be read.
    bool hessian_valid = false;
    std::vector<int> nb, jl, blk;
    std::vector<void*> owned;
    // model / weights / bounds storage
    DBuf Ahat, Bu, C, Bd, Dd, dop, Mdiag, Ndiag, Ldiag, Cwt, Mblk, Mfull, Ndense, Ldense;
    DBuf Wy, Wu, Wd, Wr, w_op, Wmin, Wmax, C_wmin, C_wmax, ry_now;
    DBuf bnd[16];
    // staging for the host-pointer step
    DBuf s_x, s_lu, s_ry, s_ru, s_d0, s_dh, s_Z, s_u0, s_st, s_it, s_yh;
    DBuf keep_q, keep_F, prof, lam, audit;
    bool lam_valid = false;     // lam holds the multipliers of the previous step (MPCQP_FLAG_WARM_DUAL)
    // SteadyKalmanFilter
    DBuf kf_K, kf_iym, kf_x, kf_y, kf_u, kf_d;
    KfParams kf{};
    bool have_kf = false;
    // MultipleShooting transcription (mpcqp_set_transcription): the stage-structured kernel of ms_bodies.h
    int transcription = MPCQP_SINGLE_SHOOTING;
    bool stage_only = false;         // nZ~ > 256: only the stage-structured kernel can take this handle (nothing is condensed)
    bool stage_rows = false;         // C_umin / C_umax vary inside a move-blocking interval: the U rows cannot be merged, the
                                     // stage-structured kernel (one U row per step) takes the handle
    bool dual_reg_given = false;     // mpcqp_dims.dual_reg > 0 (else each kernel's own default)
    DBuf ms_X, ms_defect, ms_scratch, ms_next;
};

static int dev_alloc(mpcqp_handle h, DBuf& b, size_t bytes) {
    if (b.p && b.bytes >= bytes) return MPCQP_OK;
    void* p = nullptr;
    if (bytes == 0) bytes = 8;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        g_hip_err = std::string("hipMalloc: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? MPCQP_ERR_NOMEM : MPCQP_ERR_DEVICE;
    }
    h->owned.push_back(p);
    b.p = p;
    b.bytes = bytes;
    return MPCQP_OK;
}

// give a scratch array back before the handle goes (temporaries of mpcqp_prepare's self-test)
static void dev_release(mpcqp_handle h, DBuf& b) {
    if (!b.p) return;
    for (size_t i = 0; i < h->owned.size(); ++i)
        if (h->owned[i] == b.p) { h->owned.erase(h->owned.begin() + (long)i); break; }
    (void)hipFree(b.p);
    b = DBuf{};
}

// A step enqueued on a caller's stream (mpcqp_step_device) may still read the handle's device
// arrays: every upload into them is ordered behind the end of that step (event ev_s1).
static int upload(mpcqp_handle h, DBuf& b, const void* src, size_t bytes) {
    int rc = dev_alloc(h, b, bytes);
    if (rc) return rc;
    if (h->step_timed) HIPCHK(hipStreamWaitEvent(h->stream, h->ev_s1, 0));
    HIPCHK(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, h->stream));
    return MPCQP_OK;
}

// slice of a problem-major array with `per` values per problem (NULL stays NULL)
template <class T>
static T* shard_of(T* a, int off, size_t per) { return a ? a + (size_t)off * per : nullptr; }

struct mpcqp_multi_s {
    std::vector<mpcqp_handle> h;
    std::vector<int> off, cnt;
    Dims d{};                       // whole-batch dimensions (B = total)
};

extern "C" {

#ifndef HIP_VERSION_MAJOR      // (the CPU wave emulator's stand-in runtime, tests/emu/fakehip)
#define HIP_VERSION_MAJOR 0
#define HIP_VERSION_MINOR 0
#endif
#define MPCQP_STR2(x) #x
#define MPCQP_STR(x) MPCQP_STR2(x)
// (the HIP version of the BUILD: the loader of api.py compares it with the runtime the process ends up with)
const char* mpcqp_version(void) { return "mpcqp 0.1.0 (gfx950, HIP " MPCQP_STR(HIP_VERSION_MAJOR) "." MPCQP_STR(HIP_VERSION_MINOR) ")"; }

const char* mpcqp_strerror(int code) {
    switch (code) {
        case MPCQP_OK: return "ok";
        case MPCQP_ERR_NULL: return "required pointer is NULL";
        case MPCQP_ERR_DIMS: return "dimension mismatch";
        case MPCQP_ERR_ARG: return "illegal argument value";
        case MPCQP_ERR_UNSUPPORTED: return "configuration not supported by this build";
        case MPCQP_ERR_ORDER: return "model, weights must be set before step";
        case MPCQP_ERR_DEVICE: return "HIP runtime error (see mpcqp_last_hip_error)";
        case MPCQP_ERR_NOMEM: return "out of device memory";
        default: return "unknown error code";
    }
}

const char* mpcqp_last_hip_error(void) { return g_hip_err.c_str(); }

static void layout_rows(mpcqp_handle h) {
    Dims& d = h->d;
    d.cnt_[P_BOX] = d.nZ; d.cnt_[P_U] = d.nDU; d.cnt_[P_DU] = d.nDU; d.cnt_[P_X] = d.nxh;
    d.cnt_[P_Y] = d.nY + (d.eps_host() >= 0 ? 1 : 0);         // + the row -eps <= 0 when a Ŷ group hosts it (mpcqp_types.h)
    d.cnt_[P_W] = d.nW;
    int o = 0;
    for (int g = 0; g < NGROUP; ++g) {
        d.rowoff_[g] = o;
        if ((d.gmask >> g) & 1u) o += d.cnt_[g >> 1];
    }
    d.rowoff_[NGROUP] = o;
    h->lam_valid = false;       // the row layout changed: stored multipliers no longer line up
}

int mpcqp_create(const mpcqp_dims* in, mpcqp_handle* out) {
    if (!in || !out) return MPCQP_ERR_NULL;
    *out = nullptr;
    if (in->batch < 1 || in->nxhat < 1 || in->nu < 1 || in->ny < 1 || in->nd < 0) return MPCQP_ERR_ARG;
    // validate_weights, src/controller/construct.jl:99-103
    if (in->Hp < 1 || in->Hc < 1 || in->Hc > in->Hp) return MPCQP_ERR_ARG;
    if (in->neps != 0 && in->neps != 1) return MPCQP_ERR_ARG;
    std::vector<int> nb(in->Hc, 1);
    if (in->nb) {
        int sum = 0;
        for (int i = 0; i < in->Hc; ++i) {
            if (in->nb[i] < 1) return MPCQP_ERR_ARG;   // move_blocking, construct.jl:632
            nb[i] = in->nb[i];
            sum += nb[i];
        }
        if (sum != in->Hp) return MPCQP_ERR_DIMS;
    } else {
        nb[in->Hc - 1] = in->Hp - in->Hc + 1;          // construct.jl:653-660
    }
    const int nZ = in->nu * in->Hc + in->neps;
    // nZ~ <= 64: one Cholesky row per lane (specialised kernels); above that the runtime-dims kernel
    // gives every lane several rows, as long as the problem fits the 160 KB of LDS (checked in layout_rows)
    // Beyond nZ~ = 256 (the reference has no size limit: transcription.jl:2-4) there is no condensed kernel -- the Newton
    // matrix alone would not fit the LDS.  Such handles run the SAME QP on the stage-structured kernel of ms_bodies.h
    // (cost linear in the horizon, no nZ~ limit), whatever their transcription: `stage_only`.  Nothing is condensed for
    // them (no K1 / K2 tables, no packed H~).
    const bool stage_only = nZ > 4 * WAVE;
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (in->device < 0 || in->device >= ndev) return MPCQP_ERR_ARG;
    DeviceGuard guard_(in->device);
    if (!guard_.ok) { g_hip_err = "hipSetDevice failed"; return MPCQP_ERR_DEVICE; }
    mpcqp_handle h = new (std::nothrow) mpcqp_handle_s();
    if (!h) return MPCQP_ERR_NOMEM;
    Dims& d = h->d;
    d.B = in->batch; d.nxh = in->nxhat; d.nu = in->nu; d.ny = in->ny; d.nd = in->nd;
    d.Hp = in->Hp; d.Hc = in->Hc; d.neps = in->neps;
    d.nZ = nZ; d.nDU = in->nu * in->Hc; d.nU = in->nu * in->Hp; d.nY = in->ny * in->Hp;
    d.nD = in->nd * in->Hp;
    d.nw = 0; d.nW = 0;
    d.npk = pk_size(nZ);
    d.flags = in->flags;
    d.max_iter = in->max_iter > 0 ? in->max_iter : 80;
    d.gap_tol = in->gap_tol > 0 ? in->gap_tol : 1e-12;
    d.res_tol = in->res_tol > 0 ? in->res_tol : 1e-11;
    d.dual_reg = in->dual_reg > 0 ? in->dual_reg : 1e-12;
    h->dual_reg_given = in->dual_reg > 0;
    h->stage_only = stage_only;
    d.gmask = d.neps ? 1u : 0u;                        // ϵ >= 0 is always there
    layout_rows(h);
    h->device = in->device;
    h->nb = nb;
    d.default_nb = 1;
    for (int i = 0; i < d.Hc; ++i)
        if (nb[i] != (i == d.Hc - 1 ? d.Hp - d.Hc + 1 : 1)) d.default_nb = 0;
    h->jl.assign(d.Hc + 1, 0);
    for (int i = 0; i < d.Hc; ++i) h->jl[i + 1] = h->jl[i] + nb[i];
    h->blk.assign(d.Hp, 0);
    for (int i = 0; i < d.Hc; ++i)
        for (int t = h->jl[i]; t < h->jl[i + 1]; ++t) h->blk[t] = i;
    int rc = MPCQP_OK;
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&h->ev_s0);
    if (e == hipSuccess) e = hipEventCreate(&h->ev_s1);
    if (e == hipSuccess) e = hipEventCreate(&h->ev_c0);
    if (e == hipSuccess) e = hipEventCreate(&h->ev_c1);
    if (e == hipSuccess) e = hipEventCreate(&h->ev_cm);
    if (e != hipSuccess) {
        g_hip_err = std::string("stream/event create: ") + hipGetErrorString(e);
        rc = MPCQP_ERR_DEVICE;
    }
    auto mk = [&](size_t n) -> double* {
        if (rc) return nullptr;
        DBuf b;
        rc = dev_alloc(h, b, n * sizeof(double));
        return (double*)b.p;
    };
    const size_t B = d.B;
    if (!stage_only) {
        h->m.Stab = mk(B * d.Hp * d.ny * d.nu);
        h->m.Ktab = mk(B * d.nxh * d.nY);
        h->m.Bvec = mk(B * d.nY);
        h->m.Hpk = mk(B * d.npk);
        if (d.nd > 0) h->m.Gdtab = mk(B * d.Hp * d.ny * d.nd);
    }
    if (!rc) {
        DBuf bj, bb;
        rc = dev_alloc(h, bj, (d.Hc + 1) * sizeof(int));
        if (!rc) rc = dev_alloc(h, bb, d.Hp * sizeof(int));
        if (!rc) {
            hipError_t e1 = hipMemcpy(bj.p, h->jl.data(), (d.Hc + 1) * sizeof(int), hipMemcpyHostToDevice);
            hipError_t e2 = hipMemcpy(bb.p, h->blk.data(), d.Hp * sizeof(int), hipMemcpyHostToDevice);
            if (e1 != hipSuccess || e2 != hipSuccess) { g_hip_err = "hipMemcpy(jl/blk)"; rc = MPCQP_ERR_DEVICE; }
            h->m.jl = (const int*)bj.p;
            h->m.blk = (const int*)bb.p;
        }
    }
    if (rc) { mpcqp_destroy(h); return rc; }
    *out = h;
    return MPCQP_OK;
}

int mpcqp_destroy(mpcqp_handle h) {
    if (!h) return MPCQP_ERR_NULL;
    DeviceGuard guard_(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (void* p : h->owned) (void)hipFree(p);
    if (h->ev_s0) (void)hipEventDestroy(h->ev_s0);
    if (h->ev_s1) (void)hipEventDestroy(h->ev_s1);
    if (h->ev_c0) (void)hipEventDestroy(h->ev_c0);
    if (h->ev_c1) (void)hipEventDestroy(h->ev_c1);
    if (h->ev_cm) (void)hipEventDestroy(h->ev_cm);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return MPCQP_OK;
}

int mpcqp_get_sizes(mpcqp_handle h, mpcqp_sizes* out) {
    if (!h || !out) return MPCQP_ERR_NULL;
    out->nZ = h->d.nZ; out->nDU = h->d.nDU; out->nU = h->d.nU; out->nY = h->d.nY; out->nD = h->d.nD;
    return MPCQP_OK;
}

static bool terminal_on(const Dims& d) { return (d.gmask >> (2 * P_X)) & 3u; }

// the condensed kernels keep one problem's tables in the LDS of a CU: a handle whose problem does not fit (nZ~ beyond ~165 at
// C3-like shapes; the fit test is the carve-up of the runtime-dimension kernel, which mpcqp_prepare's comparison needs) has no condensed Hessian and no condensed step -- MPCQP_ERR_UNSUPPORTED from the step / mpcqp_get, the
// MultipleShooting transcription is the way to run it
static bool aot_or_generic_other(const Dims& d) { return step_kernel_kind_other(d) == MPCQP_KERNEL_AOT; }
static bool condensed_fits(const Dims& d) { return step_lds_bytes(d) <= 160 * 1024; }

// K2 for the handle's current tables and weights, or -- when it cannot run now -- the note that Hpk is stale
static int refresh_hessian(mpcqp_handle h, hipStream_t st) {
    h->hessian_valid = false;
    if (h->have_model && h->have_weights && !h->stage_only && condensed_fits(h->d)) {
        HIPCHK(launch_hessian(h->d, h->m, st));
        h->hessian_valid = true;
    }
    return MPCQP_OK;
}
static int ensure_hessian(mpcqp_handle h, hipStream_t st) { return h->hessian_valid ? MPCQP_OK : refresh_hessian(h, st); }

static int condense(mpcqp_handle h, hipStream_t st, bool timed) {
    const Dims& d = h->d;
    if (!h->have_model || h->stage_only) return MPCQP_OK;
    const bool term = terminal_on(d);
    if (term && !h->m.exT) {
        DBuf a, b, c, x;
        int rc = dev_alloc(h, a, (size_t)d.B * d.Hc * d.nxh * d.nu * sizeof(double));
        if (!rc) rc = dev_alloc(h, b, (size_t)d.B * d.nxh * d.nxh * sizeof(double));
        if (!rc) rc = dev_alloc(h, c, (size_t)d.B * d.nxh * sizeof(double));
        if (!rc && d.nd > 0) rc = dev_alloc(h, x, (size_t)d.B * d.Hp * d.nxh * d.nd * sizeof(double));
        if (rc) return rc;
        h->m.exT = (double*)a.p; h->m.kxT = (double*)b.p; h->m.bxv = (double*)c.p;
        h->m.Xdtab = d.nd > 0 ? (double*)x.p : nullptr;
    }
    if (h->step_timed && st == h->stream) HIPCHK(hipStreamWaitEvent(st, h->ev_s1, 0));
    if (timed) HIPCHK(hipEventRecord(h->ev_c0, st));
    HIPCHK(launch_predmat(d, h->m, term, st));
    h->terminal_built = term;
    if (timed) HIPCHK(hipEventRecord(h->ev_cm, st));       // between K1 (prediction tables) and K2 (Hessian)
    { int rc = refresh_hessian(h, st); if (rc) return rc; }
    if (timed) { HIPCHK(hipEventRecord(h->ev_c1, st)); h->cond_timed = true; }
    return MPCQP_OK;
}

int mpcqp_set_model(mpcqp_handle h, const double* Ahat, const double* Bu, const double* C,
                    const double* Bd, const double* Dd, const double* dop) {
    if (!h || !Ahat || !Bu || !C) return MPCQP_ERR_NULL;
    const Dims& d = h->d;
    if (d.nd > 0 && (!Bd || !Dd)) return MPCQP_ERR_NULL;
    ON_DEVICE(h);
    const size_t B = d.B, sz = sizeof(double);
    int rc = upload(h, h->Ahat, Ahat, B * d.nxh * d.nxh * sz);
    if (!rc) rc = upload(h, h->Bu, Bu, B * d.nxh * d.nu * sz);
    if (!rc) rc = upload(h, h->C, C, B * d.ny * d.nxh * sz);
    if (!rc && d.nd > 0) rc = upload(h, h->Bd, Bd, B * d.nxh * d.nd * sz);
    if (!rc && d.nd > 0) rc = upload(h, h->Dd, Dd, B * d.ny * d.nd * sz);
    if (!rc && dop) rc = upload(h, h->dop, dop, B * d.nxh * sz);
    if (rc) return rc;
    h->m.Ahat = (const double*)h->Ahat.p; h->m.Bu = (const double*)h->Bu.p;
    h->m.C = (const double*)h->C.p;
    h->m.Bd = d.nd > 0 ? (const double*)h->Bd.p : nullptr;
    h->m.Dd = d.nd > 0 ? (const double*)h->Dd.p : nullptr;
    h->m.dop = dop ? (const double*)h->dop.p : nullptr;
    h->have_model = true;
    rc = condense(h, h->stream, true);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    return MPCQP_OK;
}

int mpcqp_set_weights(mpcqp_handle h, const double* Mdiag, const double* Ndiag,
                      const double* Ldiag, const double* Cwt) {
    if (!h || !Mdiag || !Ndiag || !Ldiag) return MPCQP_ERR_NULL;
    const Dims& d = h->d;
    if (d.neps && !Cwt) return MPCQP_ERR_NULL;
    ON_DEVICE(h);
    const size_t B = d.B, sz = sizeof(double);
    int rc = upload(h, h->Mdiag, Mdiag, B * d.nY * sz);
    if (!rc) rc = upload(h, h->Ndiag, Ndiag, B * d.nDU * sz);
    if (!rc) rc = upload(h, h->Ldiag, Ldiag, B * d.nU * sz);
    if (!rc && d.neps) rc = upload(h, h->Cwt, Cwt, B * sz);
    if (rc) return rc;
    h->m.Mdiag = (const double*)h->Mdiag.p; h->m.Ndiag = (const double*)h->Ndiag.p;
    h->m.Ldiag = (const double*)h->Ldiag.p;
    h->m.Cwt = d.neps ? (const double*)h->Cwt.p : nullptr;
    h->have_weights = true;
    { int rc = refresh_hessian(h, h->stream); if (rc) return rc; }
    HIPCHK(hipStreamSynchronize(h->stream));
    return MPCQP_OK;
}

int mpcqp_set_output_weight_blocks(mpcqp_handle h, const double* Mblk) {
    if (!h) return MPCQP_ERR_NULL;
    const Dims& d = h->d;
    if (!h->have_weights) return MPCQP_ERR_ORDER;       // N, L, C come from mpcqp_set_weights
    ON_DEVICE(h);
    if (Mblk) {
        int rc = upload(h, h->Mblk, Mblk, (size_t)d.B * d.Hp * d.ny * d.ny * sizeof(double));
        if (rc) return rc;
        h->m.Mblk = (const double*)h->Mblk.p;
    } else {
        h->m.Mblk = nullptr;
    }
    { int rc = refresh_hessian(h, h->stream); if (rc) return rc; }
    HIPCHK(hipStreamSynchronize(h->stream));
    return MPCQP_OK;
}

int mpcqp_set_dense_weights(mpcqp_handle h, const double* M_Hp, const double* N_Hc, const double* L_Hp) {
    if (!h) return MPCQP_ERR_NULL;
    Dims& d = h->d;
    if (!h->have_weights) return MPCQP_ERR_ORDER;       // the diagonals and C come from mpcqp_set_weights
    ON_DEVICE(h);
    struct { const double* src; DBuf* buf; const double** dst; size_t n; } w[3] = {
        {M_Hp, &h->Mfull, &h->m.Mfull, (size_t)d.nY}, {N_Hc, &h->Ndense, &h->m.Ndense, (size_t)d.nDU},
        {L_Hp, &h->Ldense, &h->m.Ldense, (size_t)d.nU}};
    for (auto& e : w) {
        if (e.src) {
            int rc = upload(h, *e.buf, e.src, (size_t)d.B * e.n * e.n * sizeof(double));
            if (rc) return rc;
            *e.dst = (const double*)e.buf->p;
        } else {
            *e.dst = nullptr;
        }
    }
    d.dense_w = (h->m.Mfull || h->m.Ldense) ? 1 : 0;     // (a dense N_Hc only changes H̃: any step kernel serves it)
    { int rc = refresh_hessian(h, h->stream); if (rc) return rc; }
    HIPCHK(hipStreamSynchronize(h->stream));
    return MPCQP_OK;
}

int mpcqp_set_flags(mpcqp_handle h, uint32_t flags) {
    if (!h) return MPCQP_ERR_NULL;
    const uint32_t known = MPCQP_FLAG_RY_CONSTANT | MPCQP_FLAG_COLD_START | MPCQP_FLAG_KEEP_QP | MPCQP_FLAG_WARM_DUAL | MPCQP_FLAG_NO_POLISH | MPCQP_FLAG_KEEP_ITERATE;
    if (flags & ~known) return MPCQP_ERR_ARG;
    if ((flags ^ h->d.flags) & MPCQP_FLAG_WARM_DUAL) h->lam_valid = false;
    h->d.flags = flags;
    return MPCQP_OK;
}

int mpcqp_set_custom_constraints(mpcqp_handle h, int nw, const double* Wy, const double* Wu,
                                 const double* Wd, const double* Wr, const double* w_op) {
    if (!h) return MPCQP_ERR_NULL;
    Dims& d = h->d;
    if (nw < 0) return MPCQP_ERR_ARG;
    if (nw > 0 && (!Wy || !Wu)) return MPCQP_ERR_NULL;
    ON_DEVICE(h);
    Model& m = h->m;
    d.nw = nw; d.nW = nw * (d.Hp + 1);
    d.gmask &= ~(3u << (2 * P_W));          // bounds of a previous definition are dropped
    m.Wy = m.Wu = m.Wd = m.Wr = m.w_op = nullptr;
    m.Wmin = m.Wmax = m.C_wmin = m.C_wmax = nullptr;
    if (nw > 0) {
        const size_t B = d.B, sz = sizeof(double);
        int rc = upload(h, h->Wy, Wy, B * nw * d.ny * sz);
        if (!rc) rc = upload(h, h->Wu, Wu, B * nw * d.nu * sz);
        if (!rc && Wd && d.nd > 0) rc = upload(h, h->Wd, Wd, B * nw * d.nd * sz);
        if (!rc && Wr) rc = upload(h, h->Wr, Wr, B * nw * d.ny * sz);
        if (!rc && w_op) rc = upload(h, h->w_op, w_op, B * nw * sz);
        if (rc) return rc;
        m.Wy = (const double*)h->Wy.p; m.Wu = (const double*)h->Wu.p;
        m.Wd = (Wd && d.nd > 0) ? (const double*)h->Wd.p : nullptr;
        m.Wr = Wr ? (const double*)h->Wr.p : nullptr;
        m.w_op = w_op ? (const double*)h->w_op.p : nullptr;
    }
    layout_rows(h);
    { int rc = ensure_hessian(h, h->stream); if (rc) return rc; }      // (the row groups decide whether the problem fits the LDS)
    HIPCHK(hipStreamSynchronize(h->stream));
    return MPCQP_OK;
}

int mpcqp_set_current_setpoint(mpcqp_handle h, const double* ry_now) {
    if (!h) return MPCQP_ERR_NULL;
    ON_DEVICE(h);
    if (ry_now) {
        int rc = upload(h, h->ry_now, ry_now, (size_t)h->d.B * h->d.ny * sizeof(double));
        if (rc) return rc;
        h->m.ry_now = (const double*)h->ry_now.p;
        HIPCHK(hipStreamSynchronize(h->stream));
    } else {
        h->m.ry_now = nullptr;
    }
    return MPCQP_OK;
}

int mpcqp_set_custom_bounds(mpcqp_handle h, const double* Wmin, const double* Wmax,
                            const double* C_wmin, const double* C_wmax) {
    if (!h) return MPCQP_ERR_NULL;
    Dims& d = h->d;
    if (d.nw < 1) return MPCQP_ERR_ORDER;               // mpcqp_set_custom_constraints first
    if (!d.neps && (C_wmin || C_wmax)) return MPCQP_ERR_ARG;
    ON_DEVICE(h);
    Model& m = h->m;
    const size_t n = (size_t)d.B * d.nW * sizeof(double);
    const double* src[4] = {Wmin, Wmax, C_wmin, C_wmax};
    DBuf* buf[4] = {&h->Wmin, &h->Wmax, &h->C_wmin, &h->C_wmax};
    const double* dev[4];
    for (int i = 0; i < 4; ++i) {
        dev[i] = nullptr;
        if (!src[i]) continue;
        int rc = upload(h, *buf[i], src[i], n);
        if (rc) return rc;
        dev[i] = (const double*)buf[i]->p;
    }
    m.Wmin = dev[0]; m.Wmax = dev[1]; m.C_wmin = dev[2]; m.C_wmax = dev[3];
    d.gmask &= ~(3u << (2 * P_W));
    if (m.Wmin) d.gmask |= 1u << (2 * P_W);
    if (m.Wmax) d.gmask |= 1u << (2 * P_W + 1);
    layout_rows(h);
    { int rc = ensure_hessian(h, h->stream); if (rc) return rc; }      // (the row groups decide whether the problem fits the LDS)
    HIPCHK(hipStreamSynchronize(h->stream));
    return MPCQP_OK;
}

int mpcqp_set_bounds(mpcqp_handle h, const mpcqp_bounds* bin) {
    if (!h || !bin) return MPCQP_ERR_NULL;
    Dims& d = h->d;
    if (!d.neps && (bin->C_umin || bin->C_umax || bin->C_dumin || bin->C_dumax || bin->C_ymin ||
                    bin->C_ymax || bin->c_x0min || bin->c_x0max))
        return MPCQP_ERR_ARG;   // "Cwt must be finite to set softness parameters", construct.jl:441
    // U rows of one move-blocking interval are merged into their tightest one (mpcqp_bodies.h);
    // exact as long as the softness of those rows is the same -- true for every `c_umin`/`c_umax`
    // keyword of setconstraint! (repeated per channel).  A horizon-long C_umin/C_umax vector
    // (construct.jl:454-463) that varies inside an interval leaves rows u - c_t eps <= Umax_t of different
    // slopes that no single row replaces: such a handle is served by the stage-structured kernel, which
    // keeps one U row per step (ms_bodies.h) -- if that kernel can take it (ms_unsupported at the step).
    bool varies = false;
    for (const double* cu : {bin->C_umin, bin->C_umax}) {
        if (!cu) continue;
        for (size_t b = 0; b < (size_t)d.B && !varies; ++b)
            for (int j = 0; j < d.Hc && !varies; ++j)
                for (int t = h->jl[j] + 1; t < h->jl[j + 1] && !varies; ++t)
                    for (int c = 0; c < d.nu; ++c)
                        if (cu[b * d.nU + t * d.nu + c] != cu[b * d.nU + h->jl[j] * d.nu + c]) { varies = true; break; }
    }
    h->stage_rows = varies;
    ON_DEVICE(h);
    const double* src[16] = {bin->U0min, bin->U0max, bin->DUmin, bin->DUmax, bin->Y0min, bin->Y0max,
                             bin->x0min, bin->x0max, bin->C_umin, bin->C_umax, bin->C_dumin,
                             bin->C_dumax, bin->C_ymin, bin->C_ymax, bin->c_x0min, bin->c_x0max};
    const int len[16] = {d.nU, d.nU, d.nDU, d.nDU, d.nY, d.nY, d.nxh, d.nxh,
                         d.nU, d.nU, d.nDU, d.nDU, d.nY, d.nY, d.nxh, d.nxh};
    const double* dev[16];
    for (int i = 0; i < 16; ++i) {
        dev[i] = nullptr;
        if (!src[i]) continue;
        int rc = upload(h, h->bnd[i], src[i], (size_t)d.B * len[i] * sizeof(double));
        if (rc) return rc;
        dev[i] = (const double*)h->bnd[i].p;
    }
    Model& m = h->m;
    m.U0min = dev[0]; m.U0max = dev[1]; m.DUmin = dev[2]; m.DUmax = dev[3];
    m.Y0min = dev[4]; m.Y0max = dev[5]; m.x0min = dev[6]; m.x0max = dev[7];
    m.C_umin = dev[8]; m.C_umax = dev[9]; m.C_dumin = dev[10]; m.C_dumax = dev[11];
    m.C_ymin = dev[12]; m.C_ymax = dev[13]; m.c_x0min = dev[14]; m.c_x0max = dev[15];
    uint32_t g = 0;
    // box lower: hard ΔUmin rows, and the row ϵ >= 0 unless a Ŷ group exists (it then rides there: eps_host_group)
    if (m.DUmin || (d.neps && !m.Y0min && !m.Y0max)) g |= 1u << 0;
    if (m.DUmax) g |= 1u << 1;                           // box upper
    if (m.U0min) g |= 1u << (2 * P_U);
    if (m.U0max) g |= 1u << (2 * P_U + 1);
    if (d.neps && m.DUmin && m.C_dumin) g |= 1u << (2 * P_DU);
    if (d.neps && m.DUmax && m.C_dumax) g |= 1u << (2 * P_DU + 1);
    if (m.Y0min) g |= 1u << (2 * P_Y);
    if (m.Y0max) g |= 1u << (2 * P_Y + 1);
    if (m.x0min) g |= 1u << (2 * P_X);
    if (m.x0max) g |= 1u << (2 * P_X + 1);
    g |= d.gmask & (3u << (2 * P_W));                    // custom rows: mpcqp_set_custom_bounds
    d.gmask = g;
    layout_rows(h);
    if (terminal_on(d) && !h->terminal_built && h->have_model) {
        int rc = condense(h, h->stream, false);
        if (rc) return rc;
    }
    { int rc = ensure_hessian(h, h->stream); if (rc) return rc; }      // (the row groups decide whether the problem fits the LDS)
    HIPCHK(hipStreamSynchronize(h->stream));
    return MPCQP_OK;
}

// the handle is routed to the stage-structured kernel: asked for (MultipleShooting), or the condensed kernels cannot take it
// (round 5: also a condensed problem of nZ~ <= 256 that does not fit the LDS of a CU -- nZ~ beyond ~165 at C3-like shapes -- whatever
//  its transcription: the SAME QP, solved in stage form)
static bool uses_stage_kernel(mpcqp_handle h) {
    return h->transcription == MPCQP_MULTIPLE_SHOOTING || h->stage_only || h->stage_rows || !condensed_fits(h->d);
}

// 0 when the MultipleShooting kernel takes this handle; else the reason (bit mask): 1 dense weight matrices, 2 custom
// linear constraints, 4 the stage data does not fit the 160 KB of LDS, 8 flags of the condensed kernels only
static int ms_unsupported(mpcqp_handle h) {
    int why = 0;
    if (h->m.Mfull || h->m.Ndense || h->m.Ldense) why |= 1;          // (a block-diagonal M_Hp, e.g. a terminal cost, is taken since round 5)
    if (h->m.Mblk && h->d.ny > 4 * WAVE) why |= 1;                   // (the block-weight sweep of ms_bodies.h keeps a column in four registers per lane)
    if (h->d.nw > 0) why |= 2;
    if (ms_lds_bytes(h->d, h->m) > 160 * 1024) why |= 4;
    if (h->d.flags & (MPCQP_FLAG_KEEP_QP | MPCQP_FLAG_WARM_DUAL)) why |= 8;
    return why;
}

int mpcqp_set_transcription(mpcqp_handle h, int32_t transcription) {
    if (!h) return MPCQP_ERR_NULL;
    if (transcription != MPCQP_SINGLE_SHOOTING && transcription != MPCQP_MULTIPLE_SHOOTING) return MPCQP_ERR_ARG;
    h->transcription = transcription;
    return MPCQP_OK;
}

int mpcqp_transcription_supported(mpcqp_handle h) {
    if (!h) return MPCQP_ERR_NULL;
    return uses_stage_kernel(h) ? ms_unsupported(h) : 0;
}

// shared by mpcqp_step_device (kf = false) and mpcqp_loop_device
static int step_device_impl(mpcqp_handle h, const double* xhat0, const double* lastu0, const double* Ry,
                            const double* Ru, const double* d0, const double* Dhat0, double* Ztilde, double* u0,
                            int32_t* status, int32_t* iters, double* Yhat0, void* stream,
                            const double* y0m, double* xhat0_out, int predict);

int mpcqp_step_device(mpcqp_handle h, const double* xhat0, const double* lastu0,
                      const double* Ry, const double* Ru, const double* d0, const double* Dhat0,
                      double* Ztilde, double* u0, int32_t* status, int32_t* iters,
                      double* Yhat0, void* stream) {
    return step_device_impl(h, xhat0, lastu0, Ry, Ru, d0, Dhat0, Ztilde, u0, status, iters, Yhat0, stream,
                            nullptr, nullptr, 0);
}

static int step_device_impl(mpcqp_handle h, const double* xhat0, const double* lastu0, const double* Ry,
                            const double* Ru, const double* d0, const double* Dhat0, double* Ztilde, double* u0,
                            int32_t* status, int32_t* iters, double* Yhat0, void* stream,
                            const double* y0m, double* xhat0_out, int predict) {
    if (!h || !xhat0 || !lastu0 || !Ry || !Ztilde || !u0 || !status) return MPCQP_ERR_NULL;
    const Dims& d = h->d;
    if (d.nd > 0 && (!d0 || !Dhat0)) return MPCQP_ERR_NULL;
    if (!h->have_model || !h->have_weights) return MPCQP_ERR_ORDER;
    ON_DEVICE(h);
    hipStream_t st = (hipStream_t)stream;
    StepIO io{};
    io.xhat0 = xhat0; io.lastu0 = lastu0; io.Ry = Ry; io.Ru = Ru; io.d0 = d0; io.Dhat0 = Dhat0;
    io.Z = Ztilde; io.u0 = u0; io.Yhat0 = Yhat0; io.status = status; io.iters = iters;
    if (y0m) {
        io.kf_K = h->kf.Khat; io.kf_iym = h->kf.i_ym; io.kf_nym = h->kf.nym;
        io.kf_y0m = y0m; io.xhat0_out = xhat0_out; io.kf_predict = predict;
    }
    {       // what the convergence test of every solve saw last (mpcqp_get MPCQP_GET_AUDIT)
        int rc = dev_alloc(h, h->audit, (size_t)d.B * 4 * sizeof(double));
        if (rc) return rc;
        io.audit = (double*)h->audit.p;
    }
    if (d.flags & MPCQP_FLAG_KEEP_QP) {
        int rc = dev_alloc(h, h->keep_q, (size_t)d.B * d.nZ * sizeof(double));
        if (!rc) rc = dev_alloc(h, h->keep_F, (size_t)d.B * d.nY * sizeof(double));
        if (rc) return rc;
        io.q_keep = (double*)h->keep_q.p;
        io.F_keep = (double*)h->keep_F.p;
    }
    if (d.flags & MPCQP_FLAG_WARM_DUAL) {
        const size_t nl = (size_t)d.B * (size_t)(d.nrows() > 0 ? d.nrows() : 1) * sizeof(double);
        int rc = dev_alloc(h, h->lam, nl);
        if (rc) return rc;
        io.lam_out = (double*)h->lam.p;
        io.lam_prev = h->lam_valid ? (const double*)h->lam.p : nullptr;
        h->lam_valid = true;
    }
#ifdef MPCQP_PROFILE
    {
        int rc = dev_alloc(h, h->prof, (size_t)d.B * 16 * sizeof(double));
        if (rc) return rc;
        io.prof = (double*)h->prof.p;
    }
#endif
    // (refusals first: nothing is recorded on the stream for a step that does not run)
    if (uses_stage_kernel(h) && ms_unsupported(h)) return MPCQP_ERR_UNSUPPORTED;      // (the fused Kalman loop runs on the stage-structured kernel too since round 6)
    HIPCHK(hipEventRecord(h->ev_s0, st));
    if (uses_stage_kernel(h)) {
        // the stage-structured kernel: model as equality constraints, Riccati recursion, H~ and E never formed
        MsIO ms{};
        int rc = dev_alloc(h, h->ms_X, (size_t)d.B * d.nxh * d.Hp * sizeof(double));
        if (!rc) rc = dev_alloc(h, h->ms_defect, (size_t)d.B * sizeof(double));
        if (rc) return rc;
        ms.Xhat = (double*)h->ms_X.p; ms.defect = (double*)h->ms_defect.p;
        {   // horizon-long data of the resident wavefronts when they do not fit the LDS share (ms_kernels.hip)
            int nslots = 0;
            const size_t sb = ms_scratch_bytes(d, h->m, &nslots);
            if (sb) {
                rc = dev_alloc(h, h->ms_scratch, sb);
                if (rc) return rc;
                ms.scratch = (double*)h->ms_scratch.p; ms.nslots = nslots;
                rc = dev_alloc(h, h->ms_next, sizeof(int));
                if (rc) return rc;
                ms.next = (int*)h->ms_next.p;
            }
        }
        // (dual regularisation: the handle's, default 1e-12 like the condensed kernels.  Measured on 8192 C3 controllers
        //  against the condensed kernel once the gains of the Riccati recursion are refined (ms_bodies.h: factor):
        //  delta 1e-12 / 1e-10 / 1e-9: not OPTIMAL 0 / 3 / 5, worst dU difference 2.7e-6 / 1.4e-2 / 1 (error branch), 99.9 %
        //  quantile 2e-7 / 3e-8 / 2e-7; profiles/r4/ms_delta_sweep.txt)
        Dims dm = d;
        HIPCHK(launch_ms_step(dm, h->m, io, ms, st));
    } else {
        { int rc = ensure_hessian(h, st); if (rc) return rc; }      // (skipped while the problem did not fit the LDS: see hessian_valid)
        HIPCHK(launch_step(d, h->m, io, st));
    }
    HIPCHK(hipEventRecord(h->ev_s1, st));
    h->step_timed = true;
    return MPCQP_OK;
}

int mpcqp_loop_device(mpcqp_handle h, double* xhat0, const double* y0m, const double* lastu0,
                      const double* Ry, const double* Ru, const double* d0, const double* Dhat0,
                      double* Ztilde, double* u0, int32_t* status, int32_t* iters, double* Yhat0,
                      void* stream) {
    if (!h || !y0m || !xhat0) return MPCQP_ERR_NULL;
    if (!h->have_kf) return MPCQP_ERR_ORDER;
    if (h->d.nxh > 4 * WAVE) return MPCQP_ERR_UNSUPPORTED;
    return step_device_impl(h, xhat0, lastu0, Ry, Ru, d0, Dhat0, Ztilde, u0, status, iters, Yhat0, stream,
                            y0m, xhat0, 1);
}

int mpcqp_step(mpcqp_handle h, const double* xhat0, const double* lastu0, const double* Ry,
               const double* Ru, const double* d0, const double* Dhat0, double* Ztilde,
               double* u0, int32_t* status, int32_t* iters, double* Yhat0) {
    if (!h || !xhat0 || !lastu0 || !Ry || !Ztilde || !u0 || !status) return MPCQP_ERR_NULL;
    const Dims& d = h->d;
    if (d.nd > 0 && (!d0 || !Dhat0)) return MPCQP_ERR_NULL;
    ON_DEVICE(h);
    const size_t B = d.B, sz = sizeof(double);
    const size_t nry = (d.flags & MPCQP_FLAG_RY_CONSTANT) ? d.ny : d.nY;
    int rc = upload(h, h->s_x, xhat0, B * d.nxh * sz);
    if (!rc) rc = upload(h, h->s_lu, lastu0, B * d.nu * sz);
    if (!rc) rc = upload(h, h->s_ry, Ry, B * nry * sz);
    if (!rc && Ru) rc = upload(h, h->s_ru, Ru, B * d.nU * sz);
    if (!rc && d.nd > 0) rc = upload(h, h->s_d0, d0, B * d.nd * sz);
    if (!rc && d.nd > 0) rc = upload(h, h->s_dh, Dhat0, B * d.nD * sz);
    if (!rc) rc = upload(h, h->s_Z, Ztilde, B * d.nZ * sz);
    if (!rc) rc = dev_alloc(h, h->s_u0, B * d.nu * sz);
    if (!rc) rc = dev_alloc(h, h->s_st, B * sizeof(int32_t));
    if (!rc) rc = dev_alloc(h, h->s_it, B * sizeof(int32_t));
    if (!rc && Yhat0) rc = dev_alloc(h, h->s_yh, B * d.nY * sz);
    if (rc) return rc;
    rc = mpcqp_step_device(h, (const double*)h->s_x.p, (const double*)h->s_lu.p,
                           (const double*)h->s_ry.p, Ru ? (const double*)h->s_ru.p : nullptr,
                           d.nd > 0 ? (const double*)h->s_d0.p : nullptr,
                           d.nd > 0 ? (const double*)h->s_dh.p : nullptr, (double*)h->s_Z.p,
                           (double*)h->s_u0.p, (int32_t*)h->s_st.p, (int32_t*)h->s_it.p,
                           Yhat0 ? (double*)h->s_yh.p : nullptr, h->stream);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(Ztilde, h->s_Z.p, B * d.nZ * sz, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(u0, h->s_u0.p, B * d.nu * sz, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(status, h->s_st.p, B * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    if (iters) HIPCHK(hipMemcpyAsync(iters, h->s_it.p, B * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    if (Yhat0) HIPCHK(hipMemcpyAsync(Yhat0, h->s_yh.p, B * d.nY * sz, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return MPCQP_OK;
}

int mpcqp_recondense_device(mpcqp_handle h, void* stream) {
    if (!h) return MPCQP_ERR_NULL;
    if (!h->have_model || !h->have_weights) return MPCQP_ERR_ORDER;
    ON_DEVICE(h);
    return condense(h, (hipStream_t)stream, true);
}

int mpcqp_get(mpcqp_handle h, int which, double* out) {
    if (!h || !out) return MPCQP_ERR_NULL;
    const Dims& d = h->d;
    ON_DEVICE(h);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipDeviceSynchronize());
    const size_t B = d.B;
    auto fetch = [&](const void* src, size_t n, std::vector<double>& tmp) -> int {
        tmp.resize(n);
        HIPCHK(hipMemcpy(tmp.data(), src, n * sizeof(double), hipMemcpyDeviceToHost));
        return MPCQP_OK;
    };
    std::vector<double> tmp;
    switch (which) {
        case MPCQP_GET_HESSIAN: {
            if (h->stage_only || !condensed_fits(d)) return MPCQP_ERR_UNSUPPORTED;       // (nothing is condensed for nZ~ > 256; no Hessian kernel beyond the LDS)
            if (!h->have_model || !h->have_weights) return MPCQP_ERR_ORDER;
            int rc = ensure_hessian(h, h->stream);
            if (rc) return rc;
            HIPCHK(hipStreamSynchronize(h->stream));
            rc = fetch(h->m.Hpk, B * d.npk, tmp);
            if (rc) return rc;
            for (size_t b = 0; b < B; ++b)
                for (int i = 0; i < d.nZ; ++i)
                    for (int j = 0; j <= i; ++j) {
                        double v = tmp[b * d.npk + pk(i, j)];
                        out[b * d.nZ * d.nZ + i + (size_t)d.nZ * j] = v;
                        out[b * d.nZ * d.nZ + j + (size_t)d.nZ * i] = v;
                    }
            return MPCQP_OK;
        }
        case MPCQP_GET_STEPRESP: {
            if (h->stage_only) return MPCQP_ERR_UNSUPPORTED;
            if (!h->have_model) return MPCQP_ERR_ORDER;
            int rc = fetch(h->m.Stab, B * d.Hp * d.ny * d.nu, tmp);
            if (rc) return rc;
            for (size_t b = 0; b < B; ++b)
                for (int mm = 0; mm < d.Hp; ++mm)
                    for (int a = 0; a < d.ny; ++a)
                        for (int c = 0; c < d.nu; ++c)
                            out[((b * d.Hp + mm) * d.nu + c) * d.ny + a] =
                                tmp[((b * d.Hp + mm) * d.ny + a) * d.nu + c];
            return MPCQP_OK;
        }
        case MPCQP_GET_KMAT:
            if (h->stage_only) return MPCQP_ERR_UNSUPPORTED;
            if (!h->have_model) return MPCQP_ERR_ORDER;
            HIPCHK(hipMemcpy(out, h->m.Ktab, B * d.nxh * d.nY * sizeof(double), hipMemcpyDeviceToHost));
            return MPCQP_OK;
        case MPCQP_GET_BVEC:
            if (h->stage_only) return MPCQP_ERR_UNSUPPORTED;
            if (!h->have_model) return MPCQP_ERR_ORDER;
            HIPCHK(hipMemcpy(out, h->m.Bvec, B * d.nY * sizeof(double), hipMemcpyDeviceToHost));
            return MPCQP_OK;
        case MPCQP_GET_QTILDE:
            if (!h->keep_q.p) return MPCQP_ERR_ORDER;
            HIPCHK(hipMemcpy(out, h->keep_q.p, B * d.nZ * sizeof(double), hipMemcpyDeviceToHost));
            return MPCQP_OK;
        case MPCQP_GET_FVEC:
            if (!h->keep_F.p) return MPCQP_ERR_ORDER;
            HIPCHK(hipMemcpy(out, h->keep_F.p, B * d.nY * sizeof(double), hipMemcpyDeviceToHost));
            return MPCQP_OK;
        case MPCQP_GET_XHAT_MS:
            if (!h->ms_X.p) return MPCQP_ERR_ORDER;
            HIPCHK(hipMemcpy(out, h->ms_X.p, B * d.nxh * d.Hp * sizeof(double), hipMemcpyDeviceToHost));
            return MPCQP_OK;
        case MPCQP_GET_MS_DEFECT:
            if (!h->ms_defect.p) return MPCQP_ERR_ORDER;
            HIPCHK(hipMemcpy(out, h->ms_defect.p, B * sizeof(double), hipMemcpyDeviceToHost));
            return MPCQP_OK;
        case MPCQP_GET_AUDIT:
            if (!h->audit.p) return MPCQP_ERR_ORDER;
            HIPCHK(hipMemcpy(out, h->audit.p, B * 4 * sizeof(double), hipMemcpyDeviceToHost));
            return MPCQP_OK;
#ifdef MPCQP_PROFILE
        case 99:     /* per-phase cycle counters of profiling builds (-DMPCQP_PROFILE): (16,B) */
            if (!h->prof.p) return MPCQP_ERR_ORDER;
            HIPCHK(hipMemcpy(out, h->prof.p, B * 16 * sizeof(double), hipMemcpyDeviceToHost));
            return MPCQP_OK;
#endif
        default:
            return MPCQP_ERR_ARG;
    }
}

int mpcqp_kf_set(mpcqp_handle h, const double* Khat, const int32_t* i_ym, int32_t nym) {
    if (!h || !Khat || !i_ym) return MPCQP_ERR_NULL;
    const Dims& d = h->d;
    if (nym < 1 || nym > d.ny) return MPCQP_ERR_DIMS;
    if (d.nxh > 64) return MPCQP_ERR_UNSUPPORTED;
    for (int i = 0; i < nym; ++i) {
        if (i_ym[i] < 0 || i_ym[i] >= d.ny) return MPCQP_ERR_ARG;     // validate_ym, construct.jl:190-196
        for (int j = 0; j < i; ++j)
            if (i_ym[j] == i_ym[i]) return MPCQP_ERR_ARG;
    }
    ON_DEVICE(h);
    int rc = upload(h, h->kf_K, Khat, (size_t)d.B * d.nxh * nym * sizeof(double));
    if (!rc) rc = upload(h, h->kf_iym, i_ym, (size_t)nym * sizeof(int32_t));
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    h->kf.Khat = (const double*)h->kf_K.p;
    h->kf.i_ym = (const int*)h->kf_iym.p;
    h->kf.nym = nym;
    h->have_kf = true;
    return MPCQP_OK;
}

int mpcqp_kf_correct_device(mpcqp_handle h, double* xhat0, const double* y0m, const double* d0, void* stream) {
    if (!h || !xhat0 || !y0m) return MPCQP_ERR_NULL;
    if (h->d.nd > 0 && !d0) return MPCQP_ERR_NULL;
    if (!h->have_model || !h->have_kf) return MPCQP_ERR_ORDER;
    ON_DEVICE(h);
    HIPCHK(launch_kf_correct(h->d, h->m, h->kf, xhat0, y0m, d0, (hipStream_t)stream));
    return MPCQP_OK;
}

int mpcqp_kf_predict_device(mpcqp_handle h, double* xhat0, const double* u0, const double* d0, void* stream) {
    if (!h || !xhat0 || !u0) return MPCQP_ERR_NULL;
    if (h->d.nd > 0 && !d0) return MPCQP_ERR_NULL;
    if (!h->have_model) return MPCQP_ERR_ORDER;
    if (h->d.nxh > 64) return MPCQP_ERR_UNSUPPORTED;
    ON_DEVICE(h);
    HIPCHK(launch_kf_predict(h->d, h->m, xhat0, u0, d0, (hipStream_t)stream));
    return MPCQP_OK;
}

static int kf_host(mpcqp_handle h, double* xhat0, const double* a, size_t na, const double* d0, bool correct) {
    if (!h || !xhat0 || !a) return MPCQP_ERR_NULL;
    const Dims& d = h->d;
    if (d.nd > 0 && !d0) return MPCQP_ERR_NULL;
    ON_DEVICE(h);
    const size_t B = d.B, sz = sizeof(double);
    int rc = upload(h, h->kf_x, xhat0, B * d.nxh * sz);
    if (!rc) rc = upload(h, correct ? h->kf_y : h->kf_u, a, B * na * sz);
    if (!rc && d.nd > 0) rc = upload(h, h->kf_d, d0, B * d.nd * sz);
    if (rc) return rc;
    const double* dd = d.nd > 0 ? (const double*)h->kf_d.p : nullptr;
    rc = correct ? mpcqp_kf_correct_device(h, (double*)h->kf_x.p, (const double*)h->kf_y.p, dd, h->stream)
                 : mpcqp_kf_predict_device(h, (double*)h->kf_x.p, (const double*)h->kf_u.p, dd, h->stream);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(xhat0, h->kf_x.p, B * d.nxh * sz, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return MPCQP_OK;
}

int mpcqp_kf_correct(mpcqp_handle h, double* xhat0, const double* y0m, const double* d0) {
    if (h && !h->have_kf) return MPCQP_ERR_ORDER;
    return kf_host(h, xhat0, y0m, h ? (size_t)h->kf.nym : 0, d0, true);
}

int mpcqp_kf_predict(mpcqp_handle h, double* xhat0, const double* u0, const double* d0) {
    return kf_host(h, xhat0, u0, h ? (size_t)h->d.nu : 0, d0, false);
}

static thread_local std::string g_build_err;

const char* mpcqp_last_build_error(void) { return g_build_err.c_str(); }

// One cold-started step of the first few controllers of the handle on pseudo-random states / set points, once
// with the on-demand specialisation and once with the runtime-dimension kernel: same iterates up to rounding.
static int self_test_spec_impl(mpcqp_handle h, double* worst, std::string* why, DBuf& yh, DBuf& in, DBuf& out, DBuf& sti);
static int self_test_spec(mpcqp_handle h, double* worst, std::string* why) {
    DBuf yh, in, out, sti;           // scratch of the comparison: released on every path
    const int rc = self_test_spec_impl(h, worst, why, yh, in, out, sti);
    (void)hipStreamSynchronize(h->stream);
    dev_release(h, yh); dev_release(h, in); dev_release(h, out); dev_release(h, sti);
    return rc;
}
static int self_test_spec_impl(mpcqp_handle h, double* worst, std::string* why, DBuf& yh, DBuf& in, DBuf& out, DBuf& sti) {
    Dims d = h->d;
    d.B = d.B < 8 ? d.B : 8;
    d.flags = (d.flags | MPCQP_FLAG_COLD_START) & ~(uint32_t)(MPCQP_FLAG_KEEP_QP | MPCQP_FLAG_WARM_DUAL);
    const size_t n = d.B, nry = (d.flags & MPCQP_FLAG_RY_CONSTANT) ? d.ny : d.nY;
    const size_t cnt[] = {n * d.nxh, n * d.nu, n * nry, n * (d.nd ? d.nd : 1), n * (d.nD ? d.nD : 1), n * d.nZ, n * d.nZ, n * d.nu, n * d.nu};
    // (Ŷ as well: the specialisation's predict! path is part of what is tested)
    { int rcy = dev_alloc(h, yh, n * d.nY * sizeof(double)); if (rcy) return rcy; }
    std::vector<double> host(cnt[0] + cnt[1] + cnt[2] + cnt[3] + cnt[4], 0.0);
    uint64_t lcg = 0x9E3779B97F4A7C15ull;
    auto rnd = [&] { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; return (double)(lcg >> 11) / 9007199254740992.0 * 2.0 - 1.0; };
    for (size_t i = 0; i < cnt[0]; ++i) host[i] = rnd();
    for (size_t i = 0; i < cnt[2]; ++i) host[cnt[0] + cnt[1] + i] = 2.0 * rnd();
    size_t total_in = host.size(), total_out = cnt[5] + cnt[6] + cnt[7] + cnt[8];
    int rc = dev_alloc(h, in, total_in * sizeof(double));
    if (!rc) rc = dev_alloc(h, out, total_out * sizeof(double));
    if (!rc) rc = dev_alloc(h, sti, 4 * n * sizeof(int32_t));
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(in.p, host.data(), total_in * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemsetAsync(out.p, 0, total_out * sizeof(double), h->stream));
    double* pin = (double*)in.p;
    double* pout = (double*)out.p;
    int32_t* pst = (int32_t*)sti.p;
    StepIO io{};
    io.xhat0 = pin; io.lastu0 = pin + cnt[0]; io.Ry = pin + cnt[0] + cnt[1];
    if (d.nd) { io.d0 = pin + cnt[0] + cnt[1] + cnt[2]; io.Dhat0 = io.d0 + cnt[3]; }
    io.Yhat0 = (double*)yh.p;
    io.kf_predict = 0;
    std::vector<double> z(cnt[5] + cnt[6]);
    std::vector<int32_t> s(4 * n);
    // both kernels on the same inputs with the dims `dd` (iteration cap / flags of the stage); returns the largest
    // difference of the two Z~ relative to max(1, |Z~|) (infinity on a NaN)
    auto run_pair = [&](const Dims& dd, double* rel) -> int {
        HIPCHK(hipMemsetAsync(out.p, 0, total_out * sizeof(double), h->stream));
        io.Z = pout; io.u0 = pout + cnt[5] + cnt[6]; io.status = pst; io.iters = pst + n;
        HIPCHK(launch_step_unverified_spec(dd, h->m, io, h->stream));
        io.Z = pout + cnt[5]; io.u0 = pout + cnt[5] + cnt[6] + cnt[7]; io.status = pst + 2 * n; io.iters = pst + 3 * n;
        HIPCHK(launch_step_generic(dd, h->m, io, h->stream));
        HIPCHK(hipMemcpyAsync(z.data(), pout, z.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(s.data(), pst, s.size() * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        double scale = 1.0, diff = 0.0;
        for (size_t i = 0; i < cnt[5]; ++i) {
            scale = std::fmax(scale, std::fabs(z[cnt[5] + i]));
            diff = std::fmax(diff, std::fabs(z[i] - z[cnt[5] + i]));
            if (!(z[i] == z[i])) diff = INFINITY;
        }
        *rel = diff / scale;
        return MPCQP_OK;
    };
    // Stage 1 (deterministic): the iterates after exactly 1, 2 and 3 interior-point iterations (MPCQP_FLAG_KEEP_ITERATE).
    // Same algorithm on the same data: the two kernels differ by rounding only (1e-13 measured; the starting point has
    // D~ <= 10, so the first Newton matrices are well conditioned), while a wrong entry of Phi, of a right-hand side or of a
    // row pass moves the iterate by its own size.  Iteration 1 assembles Phi over whatever the LDS held, iterations 2, 3
    // over the previous factor: the uninitialised-row class of bugs (round 3) shows in one of them.
    double worst_it = 0.0;
    int optimal_early = 0;
    for (int k = 1; k <= 3; ++k) {
        Dims dk = d;
        dk.max_iter = k;
        dk.flags |= MPCQP_FLAG_KEEP_ITERATE;
        double rel = 0.0;
        const int rc = run_pair(dk, &rel);
        if (rc) return rc;
        bool same_status = true;
        for (size_t i = 0; i < n; ++i) {
            same_status = same_status && s[i] == s[2 * n + i];
            if (s[i] == 0) ++optimal_early;        // (converged within k iterations: the optimum is compared, below)
        }
        if ((rel > 1e-8 || !same_status) && why && why->empty())
            *why = "the iterates after " + std::to_string(k) + " interior-point iteration(s) differ by " + std::to_string(rel) +
                   (same_status ? "" : " (different statuses)");
        if (!same_status) rel = INFINITY;
        worst_it = std::fmax(worst_it, rel);
    }
    // Stage 2: the complete solves -- same statuses, the same optimum.  (The iteration counts are reported, not judged:
    // an accepted or refused polish attempt moves either count by several iterations -- round 3's rule "not more than
    // max(4, half) extra iterations" rejected a correct kernel at 14 vs 6.)
    double rel_opt = 0.0;
    rc = run_pair(d, &rel_opt);
    if (rc) return rc;
    for (size_t i = 0; i < n; ++i) {
        if (s[i] != s[2 * n + i]) {
            if (why && why->empty())
                *why = "controller " + std::to_string(i) + ": status " + std::to_string(s[i]) + " after " + std::to_string(s[n + i]) +
                       " iterations, runtime-dimension kernel: status " + std::to_string(s[2 * n + i]) + " after " +
                       std::to_string(s[3 * n + i]);
            rel_opt = INFINITY;
        }
    }
    if (rel_opt > 1e-6 && why && why->empty()) *why = "the optima differ by " + std::to_string(rel_opt);
    if (getenv("MPCQP_SELFTEST_VERBOSE")) {
        fprintf(stderr, "[mpcqp] self-test %d,%d,%d,%d,%d,%d,%x: first iterates differ by %.3e, optima by %.3e; iterations",
                d.nu, d.ny, d.nxh, d.Hp, d.Hc, d.neps, d.gmask, worst_it, rel_opt);
        for (size_t i = 0; i < n; ++i) fprintf(stderr, " %d/%d", s[n + i], s[3 * n + i]);
        fprintf(stderr, " (%d early optima)\n", optimal_early);
    }
    // one figure for the caller's tolerance (1e-6): the first iterates are held to 1e-8
    *worst = std::fmax(rel_opt, worst_it * 100.0);
    return MPCQP_OK;
}

int mpcqp_prepare(mpcqp_handle h) {
    if (!h) return MPCQP_ERR_NULL;
    g_build_err.clear();
    if (uses_stage_kernel(h)) {
        // nothing to build: the stage-structured kernel is in the library.  A handle it cannot take is refused here exactly
        // as its steps are (mpcqp_step*: MPCQP_ERR_UNSUPPORTED) -- the answer is what the steps WILL do
        return ms_unsupported(h) ? MPCQP_ERR_UNSUPPORTED : MPCQP_KERNEL_MS;
    }
    { ON_DEVICE(h); int rc = ensure_hessian(h, h->stream); if (rc) return rc; }
    int kind = prepare_step(h->d, h->m, &g_build_err);
    // an on-demand kernel that has not been checked yet (fresh build, or a cache some other process filled): compare
    // it with the runtime-dimension kernel once; needs the model and weights (BatchLinMPC prepares before its first step)
    // (a handle of the small-problem kernel has its specialisation verified too: steps that ask for Ŷ or fuse the Kalman steps
    //  run on it, and so do the dense-row handles of small batches -- mpcqp_kernels.hip: small_takes)
    const bool ondemand = kind == MPCQP_KERNEL_ONDEMAND || (kind == MPCQP_KERNEL_SMALL && !aot_or_generic_other(h->d) && spec_present(h->d));
    if (ondemand && !spec_verified(h->d) && h->have_model && h->have_weights &&
        step_lds_bytes(h->d) <= 160 * 1024) {
        ON_DEVICE(h);
        double worst = 0.0;
        std::string why;
        int rc = self_test_spec(h, &worst, &why);
        if (rc) return rc;
        const char* tol_env = getenv("MPCQP_JIT_SELFTEST_TOL");       // (tests force a rejection with a negative value)
        const double tol = tol_env ? atof(tol_env) : 1e-6;
        if (worst <= tol) {
            mark_spec_verified(h->d);
            if (kind == MPCQP_KERNEL_SMALL) kind = step_kernel_kind(h->d, h->m);      // (a small batch with dense rows moves to it)
        } else {
            reject_spec(h->d);
            g_build_err = "the on-demand specialisation disagrees with the runtime-dimension kernel (relative difference " +
                          std::to_string(worst) + (why.empty() ? "" : "; " + why) + "): rejected, the runtime-dimension kernel is used";
            fprintf(stderr, "[mpcqp] %s\n", g_build_err.c_str());
            if (kind != MPCQP_KERNEL_SMALL) kind = MPCQP_KERNEL_GENERIC;
        }
    } else if (ondemand && !spec_verified(h->d)) {
        // built / found, but not compared with the runtime-dimension kernel yet (no model or weights, or the problem does
        // not fit the LDS): steps run the runtime-dimension kernel until a later mpcqp_prepare has verified the object --
        // the return value is the kind the steps WILL run on
        g_build_err = "the on-demand specialisation exists but has not been verified on this machine yet (mpcqp_prepare "
                      "after set_model and set_weights compares it with the runtime-dimension kernel)";
        if (kind != MPCQP_KERNEL_SMALL) kind = MPCQP_KERNEL_GENERIC;
    }
    return kind;
}

int mpcqp_set_iteration_limit(mpcqp_handle h, int32_t max_iter) {
    if (!h) return MPCQP_ERR_NULL;
    if (max_iter < 0) return MPCQP_ERR_ARG;
    h->d.max_iter = max_iter > 0 ? max_iter : 80;
    return MPCQP_OK;
}

int mpcqp_lds_bytes(mpcqp_handle h) {
    if (!h) return MPCQP_ERR_NULL;
    return (int)step_lds_bytes(h->d);
}

int mpcqp_kernel_kind(mpcqp_handle h) {
    if (!h) return MPCQP_ERR_NULL;
    if (uses_stage_kernel(h)) return ms_unsupported(h) ? MPCQP_ERR_UNSUPPORTED : MPCQP_KERNEL_MS;
    return step_kernel_kind(h->d, h->m);
}

int mpcqp_row_groups(mpcqp_handle h, uint32_t* row_groups) {
    if (!h || !row_groups) return MPCQP_ERR_NULL;
    *row_groups = h->d.gmask;
    return MPCQP_OK;
}

int mpcqp_prebuild(const mpcqp_dims* in, uint32_t row_groups) {
    if (!in) return MPCQP_ERR_NULL;
    if (in->nxhat < 1 || in->nu < 1 || in->ny < 1 || in->Hp < 1 || in->Hc < 1 || in->Hc > in->Hp ||
        (in->neps != 0 && in->neps != 1) || (row_groups >> NGROUP))
        return MPCQP_ERR_ARG;
    Dims d{};
    d.nxh = in->nxhat; d.nu = in->nu; d.ny = in->ny; d.nd = in->nd; d.Hp = in->Hp; d.Hc = in->Hc; d.neps = in->neps;
    d.nDU = d.nu * d.Hc; d.nZ = d.nDU + d.neps; d.nU = d.nu * d.Hp; d.nY = d.ny * d.Hp;
    d.gmask = row_groups;
    d.default_nb = 1;
    if (in->nb)
        for (int i = 0; i < d.Hc; ++i)
            if (in->nb[i] != (i == d.Hc - 1 ? d.Hp - d.Hc + 1 : 1)) d.default_nb = 0;
    g_build_err.clear();
    const int k = prebuild_step(d, &g_build_err);
    // Kernel revision 10 (round 5) moved the row -eps <= 0 into a Y group when one exists: mpcqp_set_bounds no longer sets bit 0
    // for it (C3: 0x8d -> 0x8c).  A manifest line written before that still compiles, but no handle matches the object unless
    // it really has hard dUmin rows: say so (mpcqp_last_build_error) instead of leaving the deployment on the JIT silently.
    if (k >= 0 && d.neps && (row_groups & 1u) && !(row_groups & 2u) && (row_groups & (3u << (2 * P_Y))) && g_build_err.empty())
        g_build_err = "row_groups has bit 0 (box lower) together with an output-bound group and no box upper group: since kernel "
                      "revision 10 the slack's own row rides in the output group, so only a controller with hard dUmin bounds "
                      "and no dUmax matches this object -- a pre-revision-10 manifest line? (drop bit 0: e.g. 0x8d -> 0x8c)";
    return k < 0 ? MPCQP_ERR_DEVICE : k;
}

// ---- several devices behind one handle ---------------------------------------------------------
int mpcqp_multi_create(const mpcqp_dims* in, const int32_t* device_ids, int32_t ndev, mpcqp_multi* out) {
    if (!in || !device_ids || !out) return MPCQP_ERR_NULL;
    *out = nullptr;
    if (ndev < 1 || in->batch < ndev) return MPCQP_ERR_ARG;
    mpcqp_multi mh = new (std::nothrow) mpcqp_multi_s();
    if (!mh) return MPCQP_ERR_NOMEM;
    const int B = in->batch, base = B / ndev, rem = B % ndev;
    int o = 0;
    for (int g = 0; g < ndev; ++g) {
        mpcqp_dims dg = *in;
        dg.batch = base + (g < rem ? 1 : 0);
        dg.device = device_ids[g];
        mpcqp_handle hg = nullptr;
        const int rc = mpcqp_create(&dg, &hg);
        if (rc) { mpcqp_multi_destroy(mh); return rc; }
        mh->h.push_back(hg);
        mh->off.push_back(o);
        mh->cnt.push_back(dg.batch);
        o += dg.batch;
    }
    mh->d = mh->h[0]->d;
    mh->d.B = B;
    *out = mh;
    return MPCQP_OK;
}

int mpcqp_multi_destroy(mpcqp_multi mh) {
    if (!mh) return MPCQP_ERR_NULL;
    for (mpcqp_handle hg : mh->h) (void)mpcqp_destroy(hg);
    delete mh;
    return MPCQP_OK;
}

int mpcqp_multi_ndev(mpcqp_multi mh) { return mh ? (int)mh->h.size() : MPCQP_ERR_NULL; }

mpcqp_handle mpcqp_multi_handle(mpcqp_multi mh, int32_t g) {
    return (mh && g >= 0 && g < (int)mh->h.size()) ? mh->h[g] : nullptr;
}

int mpcqp_multi_shard(mpcqp_multi mh, int32_t g, int32_t* offset, int32_t* count) {
    if (!mh || !offset || !count) return MPCQP_ERR_NULL;
    if (g < 0 || g >= (int)mh->h.size()) return MPCQP_ERR_ARG;
    *offset = mh->off[g]; *count = mh->cnt[g];
    return MPCQP_OK;
}

int mpcqp_multi_set_model(mpcqp_multi mh, const double* Ahat, const double* Bu, const double* C,
                          const double* Bd, const double* Dd, const double* dop) {
    if (!mh) return MPCQP_ERR_NULL;
    const Dims& d = mh->d;
    for (size_t g = 0; g < mh->h.size(); ++g) {
        const int o = mh->off[g];
        const int rc = mpcqp_set_model(mh->h[g], shard_of(Ahat, o, (size_t)d.nxh * d.nxh), shard_of(Bu, o, (size_t)d.nxh * d.nu),
                                       shard_of(C, o, (size_t)d.ny * d.nxh), shard_of(Bd, o, (size_t)d.nxh * d.nd),
                                       shard_of(Dd, o, (size_t)d.ny * d.nd), shard_of(dop, o, d.nxh));
        if (rc) return rc;
    }
    return MPCQP_OK;
}

int mpcqp_multi_set_weights(mpcqp_multi mh, const double* Mdiag, const double* Ndiag, const double* Ldiag,
                            const double* Cwt) {
    if (!mh) return MPCQP_ERR_NULL;
    const Dims& d = mh->d;
    for (size_t g = 0; g < mh->h.size(); ++g) {
        const int o = mh->off[g];
        const int rc = mpcqp_set_weights(mh->h[g], shard_of(Mdiag, o, d.nY), shard_of(Ndiag, o, d.nDU),
                                         shard_of(Ldiag, o, d.nU), shard_of(Cwt, o, 1));
        if (rc) return rc;
    }
    return MPCQP_OK;
}

int mpcqp_multi_set_bounds(mpcqp_multi mh, const mpcqp_bounds* b) {
    if (!mh || !b) return MPCQP_ERR_NULL;
    const Dims& d = mh->d;
    for (size_t g = 0; g < mh->h.size(); ++g) {
        const int o = mh->off[g];
        mpcqp_bounds s{};
        s.U0min = shard_of(b->U0min, o, d.nU); s.U0max = shard_of(b->U0max, o, d.nU);
        s.DUmin = shard_of(b->DUmin, o, d.nDU); s.DUmax = shard_of(b->DUmax, o, d.nDU);
        s.Y0min = shard_of(b->Y0min, o, d.nY); s.Y0max = shard_of(b->Y0max, o, d.nY);
        s.x0min = shard_of(b->x0min, o, d.nxh); s.x0max = shard_of(b->x0max, o, d.nxh);
        s.C_umin = shard_of(b->C_umin, o, d.nU); s.C_umax = shard_of(b->C_umax, o, d.nU);
        s.C_dumin = shard_of(b->C_dumin, o, d.nDU); s.C_dumax = shard_of(b->C_dumax, o, d.nDU);
        s.C_ymin = shard_of(b->C_ymin, o, d.nY); s.C_ymax = shard_of(b->C_ymax, o, d.nY);
        s.c_x0min = shard_of(b->c_x0min, o, d.nxh); s.c_x0max = shard_of(b->c_x0max, o, d.nxh);
        const int rc = mpcqp_set_bounds(mh->h[g], &s);
        if (rc) return rc;
    }
    return MPCQP_OK;
}

int mpcqp_multi_prepare(mpcqp_multi mh) {
    if (!mh) return MPCQP_ERR_NULL;
    // every shard has the same dimensions, hence the same kernel -- unless a shard's on-demand kernel was rejected on
    // its device: the runtime-dimension kernel of ANY shard is what the caller has to know about
    int kind = -1;
    for (mpcqp_handle hg : mh->h) {
        const int k = mpcqp_prepare(hg);
        if (k < 0) return k;
        if (kind < 0 || k == MPCQP_KERNEL_GENERIC) kind = k;
    }
    return kind;
}

int mpcqp_multi_step(mpcqp_multi mh, const double* xhat0, const double* lastu0, const double* Ry,
                     const double* Ru, const double* d0, const double* Dhat0, double* Ztilde,
                     double* u0, int32_t* status, int32_t* iters, double* Yhat0) {
    if (!mh || !xhat0 || !lastu0 || !Ry || !Ztilde || !u0 || !status) return MPCQP_ERR_NULL;
    const Dims& dd = mh->d;
    if (dd.nd > 0 && (!d0 || !Dhat0)) return MPCQP_ERR_NULL;
    const size_t sz = sizeof(double);
    // phase 1: uploads and the step kernel enqueued on every device's stream
    for (size_t g = 0; g < mh->h.size(); ++g) {
        mpcqp_handle h = mh->h[g];
        const Dims& d = h->d;
        const int o = mh->off[g];
        const size_t B = d.B;
        const size_t nry = (d.flags & MPCQP_FLAG_RY_CONSTANT) ? d.ny : d.nY;
        ON_DEVICE(h);
        int rc = upload(h, h->s_x, shard_of(xhat0, o, d.nxh), B * d.nxh * sz);
        if (!rc) rc = upload(h, h->s_lu, shard_of(lastu0, o, d.nu), B * d.nu * sz);
        if (!rc) rc = upload(h, h->s_ry, shard_of(Ry, o, nry), B * nry * sz);
        if (!rc && Ru) rc = upload(h, h->s_ru, shard_of(Ru, o, d.nU), B * d.nU * sz);
        if (!rc && d.nd > 0) rc = upload(h, h->s_d0, shard_of(d0, o, d.nd), B * d.nd * sz);
        if (!rc && d.nd > 0) rc = upload(h, h->s_dh, shard_of(Dhat0, o, d.nD), B * d.nD * sz);
        if (!rc) rc = upload(h, h->s_Z, shard_of(Ztilde, o, d.nZ), B * d.nZ * sz);
        if (!rc) rc = dev_alloc(h, h->s_u0, B * d.nu * sz);
        if (!rc) rc = dev_alloc(h, h->s_st, B * sizeof(int32_t));
        if (!rc) rc = dev_alloc(h, h->s_it, B * sizeof(int32_t));
        if (!rc && Yhat0) rc = dev_alloc(h, h->s_yh, B * d.nY * sz);
        if (rc) return rc;
        rc = mpcqp_step_device(h, (const double*)h->s_x.p, (const double*)h->s_lu.p, (const double*)h->s_ry.p,
                               Ru ? (const double*)h->s_ru.p : nullptr,
                               d.nd > 0 ? (const double*)h->s_d0.p : nullptr,
                               d.nd > 0 ? (const double*)h->s_dh.p : nullptr, (double*)h->s_Z.p,
                               (double*)h->s_u0.p, (int32_t*)h->s_st.p, (int32_t*)h->s_it.p,
                               Yhat0 ? (double*)h->s_yh.p : nullptr, h->stream);
        if (rc) return rc;
        // the gather: each shard's results into its slice of the caller's arrays
        HIPCHK(hipMemcpyAsync(shard_of(Ztilde, o, d.nZ), h->s_Z.p, B * d.nZ * sz, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(shard_of(u0, o, d.nu), h->s_u0.p, B * d.nu * sz, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(shard_of(status, o, 1), h->s_st.p, B * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
        if (iters) HIPCHK(hipMemcpyAsync(shard_of(iters, o, 1), h->s_it.p, B * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
        if (Yhat0) HIPCHK(hipMemcpyAsync(shard_of(Yhat0, o, d.nY), h->s_yh.p, B * d.nY * sz, hipMemcpyDeviceToHost, h->stream));
    }
    // phase 2: wait for every device
    for (mpcqp_handle h : mh->h) {
        ON_DEVICE(h);
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return MPCQP_OK;
}

int mpcqp_multi_gather_device(mpcqp_multi mh, int32_t root, const double* const* Z_shards,
                              const double* const* u0_shards, const int32_t* const* status_shards,
                              double* Z_root, double* u0_root, int32_t* status_root, void* stream) {
    if (!mh || !Z_shards || !u0_shards || !status_shards || !Z_root || !u0_root || !status_root) return MPCQP_ERR_NULL;
    if (root < 0 || root >= (int)mh->h.size()) return MPCQP_ERR_ARG;
    mpcqp_handle hr = mh->h[root];
    ON_DEVICE(hr);
    hipStream_t st = stream ? (hipStream_t)stream : hr->stream;
    const Dims& d = mh->d;
    for (size_t g = 0; g < mh->h.size(); ++g) {
        const int o = mh->off[g], src = mh->h[g]->device;
        const size_t B = mh->cnt[g];
        if (!Z_shards[g] || !u0_shards[g] || !status_shards[g]) return MPCQP_ERR_NULL;
        HIPCHK(hipMemcpyPeerAsync(shard_of(Z_root, o, d.nZ), hr->device, Z_shards[g], src, B * d.nZ * sizeof(double), st));
        HIPCHK(hipMemcpyPeerAsync(shard_of(u0_root, o, d.nu), hr->device, u0_shards[g], src, B * d.nu * sizeof(double), st));
        HIPCHK(hipMemcpyPeerAsync(shard_of(status_root, o, 1), hr->device, status_shards[g], src, B * sizeof(int32_t), st));
    }
    if (!stream) HIPCHK(hipStreamSynchronize(st));
    return MPCQP_OK;
}

int mpcqp_multi_scatter_device(mpcqp_multi mh, int32_t root, const double* xhat0_root, const double* lastu0_root,
                               const double* Ry_root, int32_t ry_rows, double* const* xhat0_shards,
                               double* const* lastu0_shards, double* const* Ry_shards, void* stream) {
    if (!mh || !xhat0_root || !lastu0_root || !Ry_root || !xhat0_shards || !lastu0_shards || !Ry_shards) return MPCQP_ERR_NULL;
    if (root < 0 || root >= (int)mh->h.size() || ry_rows <= 0) return MPCQP_ERR_ARG;
    mpcqp_handle hr = mh->h[root];
    ON_DEVICE(hr);
    hipStream_t st = stream ? (hipStream_t)stream : hr->stream;
    const Dims& d = mh->d;
    for (size_t g = 0; g < mh->h.size(); ++g) {
        const int o = mh->off[g], dst = mh->h[g]->device;
        const size_t B = mh->cnt[g];
        if (!xhat0_shards[g] || !lastu0_shards[g] || !Ry_shards[g]) return MPCQP_ERR_NULL;
        HIPCHK(hipMemcpyPeerAsync(xhat0_shards[g], dst, shard_of(xhat0_root, o, d.nxh), hr->device, B * d.nxh * sizeof(double), st));
        HIPCHK(hipMemcpyPeerAsync(lastu0_shards[g], dst, shard_of(lastu0_root, o, d.nu), hr->device, B * d.nu * sizeof(double), st));
        HIPCHK(hipMemcpyPeerAsync(Ry_shards[g], dst, shard_of(Ry_root, o, ry_rows), hr->device, B * (size_t)ry_rows * sizeof(double), st));
    }
    if (!stream) HIPCHK(hipStreamSynchronize(st));
    return MPCQP_OK;
}

static double elapsed(hipEvent_t a, hipEvent_t b, bool timed) {
    if (!timed) return -1.0;
    if (hipEventSynchronize(b) != hipSuccess) return -1.0;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, a, b) != hipSuccess) return -1.0;
    return (double)ms;
}

double mpcqp_last_step_ms(mpcqp_handle h) {
    return h ? elapsed(h->ev_s0, h->ev_s1, h->step_timed) : -1.0;
}

double mpcqp_last_condense_ms(mpcqp_handle h) {
    return h ? elapsed(h->ev_c0, h->ev_c1, h->cond_timed) : -1.0;
}

double mpcqp_last_predmat_ms(mpcqp_handle h) {
    return h ? elapsed(h->ev_c0, h->ev_cm, h->cond_timed) : -1.0;
}

}  // extern "C"
