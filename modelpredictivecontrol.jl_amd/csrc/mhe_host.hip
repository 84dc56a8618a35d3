// mhe_host.hip -- C-ABI of include/mpcqp_mhe.h: handle, device residency, launches of the batched
// linear MovingHorizonEstimator.  No CPU fallback: every compute entry point needs a HIP device.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/mpcqp_mhe.h"
#include "mhe_launch.h"
#include "mpcqp_hostutil.h"

using namespace mpcqp;

struct mpcqp_mhe_s {
    mhe::Dims d{};
    mhe::Args a{};
    mhe::Raw raw{};
    int device = 0;
    uint32_t flags = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false, have_model = false, have_init = false;
    int Nk = 0;
    std::vector<void*> owned;
    // device arrays
    double *lastu = nullptr, *P0 = nullptr, *Pout = nullptr;
    double* raw_own[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // device copies of set_model's arrays
    double* bnd[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    double* bndw[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};    // window-long bound arrays (CLS_L)
    double* sft[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    double* sftw[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};    // window-long softness arrays (CLS_C)
    double* Cwt = nullptr;
    bool soft = false;       // finite Cwt: a slack variable exists
    double *s_y = nullptr, *s_u = nullptr, *s_d = nullptr;      // staging of the host-pointer entry points
    size_t scratch_bytes = 0;
};

static int dalloc(mpcqp_mhe h, void** p, size_t bytes) {
    if (bytes == 0) bytes = 8;
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {
        g_hip_err = std::string("hipMalloc: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? MPCQP_ERR_NOMEM : MPCQP_ERR_DEVICE;
    }
    h->owned.push_back(*p);
    return MPCQP_OK;
}
template <class T>
static int dalloc_t(mpcqp_mhe h, T** p, size_t n) { return dalloc(h, (void**)p, n * sizeof(T)); }

static int up(mpcqp_mhe h, double* dst, const double* src, size_t n) {
    HIPCHK(hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    return MPCQP_OK;
}

static int ensure_scratch(mpcqp_mhe h) {
    const mhe::SlotMap sm = mhe::slot_map(h->d.NX, h->d.He, h->d.cls);
    h->d.nslot = sm.total;
    const size_t need = (size_t)h->d.nwaves * mhe::wave_scratch_doubles(h->d.NX, sm.total) * sizeof(double);
    if (need <= h->scratch_bytes) return MPCQP_OK;
    void* p = nullptr;
    int rc = dalloc(h, &p, need);
    if (rc) return rc;
    h->a.scratch = (double*)p;
    h->scratch_bytes = need;
    return MPCQP_OK;
}

// one estimation period on the device: window bookkeeping (add_data_windows!, execute.jl:497-548),
// arrival covariance correction when the window moves (current form, execute.jl:727-745), solve
static int solve_period(mpcqp_mhe h, const double* y_dev, const double* d_dev, const double* u_dev) {
    mhe::Dims& d = h->d;
    bool moving = false;
    if (h->Nk < d.He) {
        ++h->Nk;
    } else {
        d.hy = (d.hy + 1) % d.He;
        d.hd = (d.hd + 1) % (d.He + 1);
        moving = true;
    }
    d.N = h->Nk;
    int rc = ensure_scratch(h);
    if (rc) return rc;
    if (d.direct && moving) HIPCHK(mhe::launch_cov(d, h->a, 1, nullptr, nullptr, h->stream));
    mhe::Args a = h->a;
    a.y0m_new = y_dev;
    a.d0_new = d_dev;
    a.u0_new = u_dev;
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    HIPCHK(mhe::launch_step(d, a, h->stream));
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    return MPCQP_OK;
}

static int count_failed(mpcqp_mhe h) {
    std::vector<int32_t> st(h->d.B);
    HIPCHK(hipMemcpyAsync(st.data(), h->a.status, st.size() * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    int bad = 0;
    for (int32_t s : st) bad += s != 0;
    return bad;
}

extern "C" {

int mpcqp_mhe_create(const mpcqp_mhe_dims* in, mpcqp_mhe* out) {
    if (!in || !out) return MPCQP_ERR_NULL;
    *out = nullptr;
    if (in->batch < 1 || in->nxhat < 1 || in->nu < 0 || in->nym < 1 || in->nd < 0 || in->He < 1) return MPCQP_ERR_DIMS;
    if (in->flags & ~MPCQP_MHE_KEEP_WINDOWS) return MPCQP_ERR_ARG;
    if (in->nxhat > mhe::RL || in->nym > mhe::RL) return MPCQP_ERR_UNSUPPORTED;
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (in->device < 0 || in->device >= ndev) return MPCQP_ERR_ARG;
    DeviceGuard guard_(in->device);
    if (!guard_.ok) { g_hip_err = "hipSetDevice failed"; return MPCQP_ERR_DEVICE; }
    mpcqp_mhe h = new (std::nothrow) mpcqp_mhe_s();
    if (!h) return MPCQP_ERR_NOMEM;
    mhe::Dims& d = h->d;
    d.B = in->batch; d.nx = in->nxhat; d.nu = in->nu; d.nym = in->nym; d.nd = in->nd; d.He = in->He;
    d.direct = in->direct ? 1 : 0;
    const int nmax = in->nxhat > in->nym ? in->nxhat : in->nym;
    d.NX = 4 * ((nmax + 3) / 4);
    d.N = 0; d.hy = 0; d.hd = 0; d.cls = 0;
    d.max_iter = in->max_iter > 0 ? in->max_iter : 80;
    d.gap_tol = in->gap_tol > 0 ? in->gap_tol : 1e-12;
    d.res_tol = in->res_tol > 0 ? in->res_tol : 1e-11;
    // (dual regularisation: rows held at D~ = 1/δ; 1e-10 -- two soft-bound families of the randomised sweeps sat on the noise floor of
    // the block recursion with 1e-12, errors 1e-5..7e-5 -- δ vanishes from the converged solution)
    d.dual_reg = in->dual_reg > 0 ? in->dual_reg : 1e-10;
    d.nwaves = mhe::waves_for(in->device, d.B, d.NX);
    d.cst_stride = mhe::cst_map(d.NX, d.nu, d.nd).stride;
    d.opt = getenv("MPCQP_MHE_OPT") ? (uint32_t)atoi(getenv("MPCQP_MHE_OPT")) : 0u;
    h->device = in->device;
    h->flags = in->flags;
    int rc = MPCQP_OK;
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&h->ev0);
    if (e == hipSuccess) e = hipEventCreate(&h->ev1);
    if (e != hipSuccess) { g_hip_err = std::string("stream/event create: ") + hipGetErrorString(e); rc = MPCQP_ERR_DEVICE; }
    const size_t B = d.B, nx = d.nx, nu = d.nu, nym = d.nym, nd = d.nd, He = d.He;
    auto mk = [&](double** p, size_t n) { if (!rc) rc = dalloc_t(h, p, n); };
    mk(&h->a.cst, B * d.cst_stride);
    mk(&h->a.P, B * d.NX * mhe::RL);
    mk(&h->a.Pi2, B * d.NX * mhe::RL);
    mk(&h->a.Y0m, B * He * nym); mk(&h->a.U0, B * He * nu); mk(&h->a.D0, B * (He + 1) * nd); mk(&h->a.X0old, B * He * nx);
    mk(&h->a.xhat0, B * nx);
    mk(&h->a.Zt, B * (nx + He * nx));
    if (in->flags & MPCQP_MHE_KEEP_WINDOWS) { mk(&h->a.Vhat, B * He * nym); mk(&h->a.Xhat, B * He * nx); }
    mk(&h->lastu, B * nu); mk(&h->P0, B * nx * nx); mk(&h->Pout, B * nx * nx);
    mk(&h->s_y, B * nym); mk(&h->s_u, B * nu); mk(&h->s_d, B * nd);
    if (!rc) rc = dalloc_t(h, &h->a.status, B);
    if (!rc) rc = dalloc_t(h, &h->a.iters, B);
    if (rc) { mpcqp_mhe_destroy(h); return rc; }
    *out = h;
    return MPCQP_OK;
}

int mpcqp_mhe_destroy(mpcqp_mhe h) {
    if (!h) return MPCQP_OK;
    DeviceGuard guard_(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (void* p : h->owned) (void)hipFree(p);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return MPCQP_OK;
}

int mpcqp_mhe_set_model(mpcqp_mhe h, const double* Ahat, const double* Bhu, const double* Chm, const double* Bhd,
                        const double* Dhdm, const double* fx, const double* Qhat, const double* Rhat) {
    if (!h || !Ahat || !Chm || !Qhat || !Rhat) return MPCQP_ERR_NULL;
    const mhe::Dims& d = h->d;
    if ((d.nu > 0 && !Bhu) || (d.nd > 0 && (!Bhd || !Dhdm))) return MPCQP_ERR_NULL;
    ON_DEVICE(h);
    const size_t B = d.B, nx = d.nx, nu = d.nu, nym = d.nym, nd = d.nd;
    int rc = MPCQP_OK;
    // (the sizes are fixed by the handle's dimensions: a second set_model -- setmodel! -- reuses the first one's arrays)
    int k = 0;
    auto put = [&](const double** slot, const double* src, size_t n) {
        double*& keep = h->raw_own[k++];
        if (rc || !src || n == 0) { if (!src) *slot = nullptr; return; }
        if (!keep) rc = dalloc_t(h, &keep, n);
        if (!rc) rc = up(h, keep, src, n);
        *slot = keep;
    };
    put(&h->raw.Ahat, Ahat, B * nx * nx); put(&h->raw.Bu, Bhu, B * nx * nu); put(&h->raw.Cm, Chm, B * nym * nx);
    put(&h->raw.Bd, Bhd, B * nx * nd); put(&h->raw.Ddm, Dhdm, B * nym * nd); put(&h->raw.fx, fx, B * nx);
    put(&h->raw.Q, Qhat, B * nx * nx); put(&h->raw.R, Rhat, B * nym * nym);
    if (rc) return rc;
    HIPCHK(mhe::launch_setup(d, h->raw, h->a.cst, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->have_model = true;
    return MPCQP_OK;
}

int mpcqp_mhe_set_bounds(mpcqp_mhe h, const double* xmin, const double* xmax, const double* wmin, const double* wmax,
                         const double* vmin, const double* vmax) {
    if (!h) return MPCQP_ERR_NULL;
    ON_DEVICE(h);
    mhe::Dims& d = h->d;
    const double* src[6] = {xmin, xmax, wmin, wmax, vmin, vmax};
    const int n[6] = {d.nx, d.nx, d.nx, d.nx, d.nym, d.nym};
    const uint32_t bit[6] = {mhe::CLS_X, mhe::CLS_X, mhe::CLS_W, mhe::CLS_W, mhe::CLS_V, mhe::CLS_V};
    const double** dst[6] = {&h->a.xmin, &h->a.xmax, &h->a.wmin, &h->a.wmax, &h->a.vmin, &h->a.vmax};
    uint32_t cls = 0;
    std::vector<double> buf((size_t)d.B * mhe::RL);
    for (int k = 0; k < 6; ++k) {
        const bool lower = (k % 2) == 0;
        bool any = false;
        for (size_t b = 0; b < (size_t)d.B; ++b)
            for (int r = 0; r < mhe::RL; ++r) {
                double v = (src[k] && r < n[k]) ? src[k][b * n[k] + r] : (lower ? -INFINITY : INFINITY);
                if (v != v) return MPCQP_ERR_ARG;
                if (std::isinf(v) || std::fabs(v) >= BIG) v = lower ? -BIG : BIG; else any = true;
                buf[b * mhe::RL + r] = v;
            }
        if (any) {
            if (!h->bnd[k]) { int rc = dalloc_t(h, &h->bnd[k], buf.size()); if (rc) return rc; }
            int rc = up(h, h->bnd[k], buf.data(), buf.size());
            if (rc) return rc;
            HIPCHK(hipStreamSynchronize(h->stream));
            *dst[k] = h->bnd[k];
            cls |= bit[k];
        } else {
            *dst[k] = nullptr;
        }
    }
    d.cls = cls | (d.cls & (mhe::CLS_S | mhe::CLS_C));     // (the softness set earlier stays: CLS_C says its arrays are window-long)
    return MPCQP_OK;
}

int mpcqp_mhe_set_bounds_window(mpcqp_mhe h, const double* Xmin, const double* Xmax, const double* Wmin, const double* Wmax,
                                const double* Vmin, const double* Vmax) {
    if (!h) return MPCQP_ERR_NULL;
    ON_DEVICE(h);
    mhe::Dims& d = h->d;
    const double* src[6] = {Xmin, Xmax, Wmin, Wmax, Vmin, Vmax};
    const int n[6] = {d.nx, d.nx, d.nx, d.nx, d.nym, d.nym};
    const int nblk[6] = {d.He + 1, d.He + 1, d.He, d.He, d.He, d.He};
    const uint32_t bit[6] = {mhe::CLS_X, mhe::CLS_X, mhe::CLS_W, mhe::CLS_W, mhe::CLS_V, mhe::CLS_V};
    const double** dst[6] = {&h->a.xmin, &h->a.xmax, &h->a.wmin, &h->a.wmax, &h->a.vmin, &h->a.vmax};
    uint32_t cls = 0;
    for (int k = 0; k < 6; ++k) {
        const bool lower = (k % 2) == 0;
        bool any = false;
        std::vector<double> buf((size_t)d.B * nblk[k] * mhe::RL);
        for (size_t b = 0; b < (size_t)d.B; ++b)
            for (int j = 0; j < nblk[k]; ++j)
                for (int r = 0; r < mhe::RL; ++r) {
                    double v = (src[k] && r < n[k]) ? src[k][(b * nblk[k] + j) * n[k] + r] : (lower ? -INFINITY : INFINITY);
                    if (v != v) return MPCQP_ERR_ARG;
                    if (std::isinf(v) || std::fabs(v) >= BIG) v = lower ? -BIG : BIG; else any = true;
                    buf[(b * nblk[k] + j) * mhe::RL + r] = v;
                }
        if (any) {
            if (!h->bndw[k]) { int rc = dalloc_t(h, &h->bndw[k], buf.size()); if (rc) return rc; }
            int rc = up(h, h->bndw[k], buf.data(), buf.size());
            if (rc) return rc;
            HIPCHK(hipStreamSynchronize(h->stream));
            *dst[k] = h->bndw[k];
            cls |= bit[k];
        } else {
            *dst[k] = nullptr;
        }
    }
    d.cls = cls | (d.cls & (mhe::CLS_S | mhe::CLS_C)) | mhe::CLS_L;
    return MPCQP_OK;
}

static int set_cwt(mpcqp_mhe h, const double* Cwt) {        // finite weights of ε², the slack's output array
    mhe::Dims& d = h->d;
    for (size_t b = 0; b < (size_t)d.B; ++b)
        if (!(Cwt[b] >= 0.0) || std::isinf(Cwt[b])) return MPCQP_ERR_ARG;
    if (!h->Cwt) { int rc = dalloc_t(h, &h->Cwt, (size_t)d.B); if (rc) return rc; }
    int rc = up(h, h->Cwt, Cwt, (size_t)d.B);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    h->a.Cwt = h->Cwt;
    if (!h->a.eps_out) { rc = dalloc_t(h, &h->a.eps_out, (size_t)d.B); if (rc) return rc; }
    return MPCQP_OK;
}

int mpcqp_mhe_set_softness(mpcqp_mhe h, const double* Cwt, const double* c_xmin, const double* c_xmax, const double* c_wmin,
                           const double* c_wmax, const double* c_vmin, const double* c_vmax) {
    if (!h) return MPCQP_ERR_NULL;
    ON_DEVICE(h);
    mhe::Dims& d = h->d;
    const double* src[6] = {c_xmin, c_xmax, c_wmin, c_wmax, c_vmin, c_vmax};
    const int n[6] = {d.nx, d.nx, d.nx, d.nx, d.nym, d.nym};
    const double** dst[6] = {&h->a.cxmin, &h->a.cxmax, &h->a.cwmin, &h->a.cwmax, &h->a.cvmin, &h->a.cvmax};
    bool any_c = false;
    for (int k = 0; k < 6; ++k)
        if (src[k])
            for (size_t i = 0; i < (size_t)d.B * n[k]; ++i) {
                if (!(src[k][i] >= 0.0) || std::isinf(src[k][i])) return MPCQP_ERR_ARG;      // softness is >= 0 and finite
                any_c = any_c || src[k][i] > 0.0;
            }
    if (!Cwt) {                       // Cwt = Inf: hard constraints only
        if (any_c) return MPCQP_ERR_ARG;     // (the reference: "Cwt must be finite to set softness parameters")
        d.cls &= ~(mhe::CLS_S | mhe::CLS_C);
        h->soft = false;
        for (int k = 0; k < 6; ++k) *dst[k] = nullptr;
        return MPCQP_OK;
    }
    int rc = set_cwt(h, Cwt);
    if (rc) return rc;
    std::vector<double> buf((size_t)d.B * mhe::RL);
    for (int k = 0; k < 6; ++k) {
        if (!src[k]) { *dst[k] = nullptr; continue; }
        for (size_t b = 0; b < (size_t)d.B; ++b)
            for (int r = 0; r < mhe::RL; ++r) buf[b * mhe::RL + r] = r < n[k] ? src[k][b * n[k] + r] : 0.0;
        if (!h->sft[k]) { rc = dalloc_t(h, &h->sft[k], buf.size()); if (rc) return rc; }
        rc = up(h, h->sft[k], buf.data(), buf.size());
        if (rc) return rc;
        HIPCHK(hipStreamSynchronize(h->stream));
        *dst[k] = h->sft[k];
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    d.cls = (d.cls | mhe::CLS_S) & ~mhe::CLS_C;
    h->soft = true;
    return MPCQP_OK;
}

int mpcqp_mhe_set_softness_window(mpcqp_mhe h, const double* Cwt, const double* C_xmin, const double* C_xmax, const double* C_wmin,
                                  const double* C_wmax, const double* C_vmin, const double* C_vmax) {
    if (!h) return MPCQP_ERR_NULL;
    if (!Cwt) return MPCQP_ERR_ARG;          // (the reference: "Cwt must be finite to set softness parameters")
    ON_DEVICE(h);
    mhe::Dims& d = h->d;
    const double* src[6] = {C_xmin, C_xmax, C_wmin, C_wmax, C_vmin, C_vmax};
    const int n[6] = {d.nx, d.nx, d.nx, d.nx, d.nym, d.nym};
    const int nblk[6] = {d.He + 1, d.He + 1, d.He, d.He, d.He, d.He};
    const double** dst[6] = {&h->a.cxmin, &h->a.cxmax, &h->a.cwmin, &h->a.cwmax, &h->a.cvmin, &h->a.cvmax};
    for (int k = 0; k < 6; ++k)
        if (src[k])
            for (size_t i = 0; i < (size_t)d.B * nblk[k] * n[k]; ++i)
                if (!(src[k][i] >= 0.0) || std::isinf(src[k][i])) return MPCQP_ERR_ARG;      // softness is >= 0 and finite
    int rc = set_cwt(h, Cwt);
    if (rc) return rc;
    for (int k = 0; k < 6; ++k) {
        if (!src[k]) { *dst[k] = nullptr; continue; }
        std::vector<double> buf((size_t)d.B * nblk[k] * mhe::RL);
        for (size_t b = 0; b < (size_t)d.B; ++b)
            for (int j = 0; j < nblk[k]; ++j)
                for (int r = 0; r < mhe::RL; ++r)
                    buf[(b * nblk[k] + j) * mhe::RL + r] = r < n[k] ? src[k][(b * nblk[k] + j) * n[k] + r] : 0.0;
        if (!h->sftw[k]) { rc = dalloc_t(h, &h->sftw[k], buf.size()); if (rc) return rc; }
        rc = up(h, h->sftw[k], buf.data(), buf.size());
        if (rc) return rc;
        HIPCHK(hipStreamSynchronize(h->stream));
        *dst[k] = h->sftw[k];
    }
    d.cls |= mhe::CLS_S | mhe::CLS_C;
    h->soft = true;
    return MPCQP_OK;
}

int mpcqp_mhe_init(mpcqp_mhe h, const double* xhat0, const double* P0, const double* d0_prev, const double* lastu0) {
    if (!h || !P0) return MPCQP_ERR_NULL;
    if (!h->have_model) return MPCQP_ERR_ORDER;
    ON_DEVICE(h);
    mhe::Dims& d = h->d;
    const size_t B = d.B, nx = d.nx, nu = d.nu, nd = d.nd, He = d.He;
    HIPCHK(hipMemsetAsync(h->a.xhat0, 0, B * nx * sizeof(double), h->stream));
    if (nd) HIPCHK(hipMemsetAsync(h->a.D0, 0, B * (He + 1) * nd * sizeof(double), h->stream));
    if (nu) HIPCHK(hipMemsetAsync(h->lastu, 0, B * nu * sizeof(double), h->stream));
    int rc = MPCQP_OK;
    if (xhat0) rc = up(h, h->a.xhat0, xhat0, B * nx);
    if (!rc && lastu0 && nu) rc = up(h, h->lastu, lastu0, B * nu);
    if (!rc && d0_prev && nd) {       // d0(-1): entry 0 of every D0 window
        std::vector<double> w(B * (He + 1) * nd, 0.0);
        for (size_t b = 0; b < B; ++b) std::memcpy(&w[b * (He + 1) * nd], d0_prev + b * nd, nd * sizeof(double));
        rc = up(h, h->a.D0, w.data(), w.size());
        if (!rc) HIPCHK(hipStreamSynchronize(h->stream));
    }
    if (!rc) rc = up(h, h->P0, P0, B * nx * nx);
    if (rc) return rc;
    d.hy = d.hd = 0;
    h->Nk = 0;
    HIPCHK(mhe::launch_cov(d, h->a, 4, h->P0, nullptr, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->have_init = true;
    return MPCQP_OK;
}

int mpcqp_mhe_set_state(mpcqp_mhe h, const double* xhat0) {
    if (!h || !xhat0) return MPCQP_ERR_NULL;
    if (!h->have_init) return MPCQP_ERR_ORDER;
    ON_DEVICE(h);
    const int rc = up(h, h->a.xhat0, xhat0, (size_t)h->d.B * h->d.nx);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    return MPCQP_OK;
}

int mpcqp_mhe_shift_windows(mpcqp_mhe h, const double* dy0m, const double* du0, const double* dd0, const double* dx0) {
    if (!h) return MPCQP_ERR_NULL;
    if (!h->have_init) return MPCQP_ERR_ORDER;
    ON_DEVICE(h);
    const mhe::Dims& d = h->d;
    const size_t B = d.B;
    HIPCHK(hipStreamSynchronize(h->stream));
    // (a rare host-side operation: the windows come down, take the offsets and go back)
    auto shift = [&](double* dev, size_t blocks, size_t n, const double* delta) -> int {
        if (!delta || !dev || n == 0 || blocks == 0) return MPCQP_OK;
        std::vector<double> w(B * blocks * n);
        HIPCHK(hipMemcpy(w.data(), dev, w.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (size_t b = 0; b < B; ++b)
            for (size_t j = 0; j < blocks; ++j)
                for (size_t i = 0; i < n; ++i) w[(b * blocks + j) * n + i] += delta[b * n + i];
        HIPCHK(hipMemcpy(dev, w.data(), w.size() * sizeof(double), hipMemcpyHostToDevice));
        return MPCQP_OK;
    };
    int rc = shift(h->a.Y0m, d.He, d.nym, dy0m);
    if (!rc) rc = shift(h->a.U0, d.He, d.nu, du0);
    if (!rc) rc = shift(h->lastu, 1, d.nu, du0);
    if (!rc) rc = shift(h->a.D0, d.He + 1, d.nd, dd0);
    if (!rc) rc = shift(h->a.X0old, d.He, d.nx, dx0);
    if (!rc) rc = shift(h->a.xhat0, 1, d.nx, dx0);
    if (!rc && h->a.Zt && dx0) {          // x̂0arr of the last solve (first nx̂ entries of Z̃ per estimator)
        std::vector<double> z(B * (size_t)(d.nx + d.He * d.nx));
        HIPCHK(hipMemcpy(z.data(), h->a.Zt, z.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (size_t b = 0; b < B; ++b)
            for (int i = 0; i < d.nx; ++i) z[b * (size_t)(d.nx + d.He * d.nx) + i] += dx0[b * d.nx + i];
        HIPCHK(hipMemcpy(h->a.Zt, z.data(), z.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    return rc;
}

int mpcqp_mhe_prepare_device(mpcqp_mhe h, const double* y0m_dev, const double* d0_dev) {
    if (!h || !y0m_dev) return MPCQP_ERR_NULL;
    if (h->d.nd > 0 && !d0_dev) return MPCQP_ERR_NULL;
    if (!h->have_init) return MPCQP_ERR_ORDER;
    if (!h->d.direct) return MPCQP_OK;
    ON_DEVICE(h);
    return solve_period(h, y0m_dev, d0_dev, h->lastu);
}

int mpcqp_mhe_update_device(mpcqp_mhe h, const double* u0_dev, const double* y0m_dev, const double* d0_dev) {
    if (!h) return MPCQP_ERR_NULL;
    if (h->d.nu > 0 && !u0_dev) return MPCQP_ERR_NULL;
    if (!h->have_init) return MPCQP_ERR_ORDER;
    ON_DEVICE(h);
    const mhe::Dims& d = h->d;
    if (!d.direct) {
        if (!y0m_dev || (d.nd > 0 && !d0_dev)) return MPCQP_ERR_NULL;
        int rc = solve_period(h, y0m_dev, d0_dev, u0_dev);
        if (rc) return rc;
    }
    // update_cov! (execute.jl:755-781): once the window is full, the arrival covariance advances by one
    // KalmanFilter period (prediction only in the current form: it was corrected in preparestate!)
    if (h->Nk == d.He) HIPCHK(mhe::launch_cov(d, h->a, d.direct ? 2 : 3, nullptr, nullptr, h->stream));
    if (d.nu > 0)
        HIPCHK(hipMemcpyAsync(h->lastu, u0_dev, (size_t)d.B * d.nu * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    return MPCQP_OK;
}

int mpcqp_mhe_sync(mpcqp_mhe h) {
    if (!h) return MPCQP_ERR_NULL;
    ON_DEVICE(h);
    HIPCHK(hipStreamSynchronize(h->stream));
    return MPCQP_OK;
}

int mpcqp_mhe_prepare(mpcqp_mhe h, const double* y0m, const double* d0) {
    if (!h || !y0m) return MPCQP_ERR_NULL;
    if (h->d.nd > 0 && !d0) return MPCQP_ERR_NULL;
    if (!h->have_init) return MPCQP_ERR_ORDER;
    if (!h->d.direct) return MPCQP_OK;
    ON_DEVICE(h);
    int rc = up(h, h->s_y, y0m, (size_t)h->d.B * h->d.nym);
    if (!rc && h->d.nd) rc = up(h, h->s_d, d0, (size_t)h->d.B * h->d.nd);
    if (!rc) rc = mpcqp_mhe_prepare_device(h, h->s_y, h->s_d);
    if (rc) return rc;
    return count_failed(h);
}

int mpcqp_mhe_update(mpcqp_mhe h, const double* u0, const double* y0m, const double* d0) {
    if (!h) return MPCQP_ERR_NULL;
    const mhe::Dims& d = h->d;
    if (d.nu > 0 && !u0) return MPCQP_ERR_NULL;
    if (!d.direct && (!y0m || (d.nd > 0 && !d0))) return MPCQP_ERR_NULL;
    if (!h->have_init) return MPCQP_ERR_ORDER;
    ON_DEVICE(h);
    int rc = MPCQP_OK;
    if (d.nu) rc = up(h, h->s_u, u0, (size_t)d.B * d.nu);
    if (!rc && !d.direct) rc = up(h, h->s_y, y0m, (size_t)d.B * d.nym);
    if (!rc && !d.direct && d.nd) rc = up(h, h->s_d, d0, (size_t)d.B * d.nd);
    if (!rc) rc = mpcqp_mhe_update_device(h, h->s_u, h->s_y, h->s_d);
    if (rc) return rc;
    if (!d.direct) return count_failed(h);
    HIPCHK(hipStreamSynchronize(h->stream));
    return MPCQP_OK;
}

void* mpcqp_mhe_device_ptr(mpcqp_mhe h, int what) {
    if (!h) return nullptr;
    switch (what) {
        case MPCQP_MHE_XHAT0: return h->a.xhat0;
        case MPCQP_MHE_ZTILDE: return h->a.Zt;
        case MPCQP_MHE_STATUS: return h->a.status;
        case MPCQP_MHE_ITERS: return h->a.iters;
        case MPCQP_MHE_VHAT: return h->a.Vhat;
        case MPCQP_MHE_XHATWIN: return h->a.Xhat;
        default: return nullptr;
    }
}

int mpcqp_mhe_get(mpcqp_mhe h, int what, void* out) {
    if (!h || !out) return MPCQP_ERR_NULL;
    ON_DEVICE(h);
    const mhe::Dims& d = h->d;
    const size_t B = d.B, nx = d.nx, He = d.He;
    const void* src = nullptr;
    size_t bytes = 0;
    switch (what) {
        case MPCQP_MHE_XHAT0: src = h->a.xhat0; bytes = B * nx * 8; break;
        case MPCQP_MHE_ZTILDE: src = h->a.Zt; bytes = B * (nx + He * nx) * 8; break;
        case MPCQP_MHE_STATUS: src = h->a.status; bytes = B * 4; break;
        case MPCQP_MHE_ITERS: src = h->a.iters; bytes = B * 4; break;
        case MPCQP_MHE_VHAT: src = h->a.Vhat; bytes = B * He * d.nym * 8; break;
        case MPCQP_MHE_XHATWIN: src = h->a.Xhat; bytes = B * He * nx * 8; break;
        case MPCQP_MHE_EPSILON: src = h->a.eps_out; bytes = B * 8; break;
        case MPCQP_MHE_PBAR: {
            // the row-lane array back to (nx̂,nx̂,B): a covariance launch with no update writes the ABI copy
            HIPCHK(mhe::launch_cov(d, h->a, 0, nullptr, h->Pout, h->stream));
            src = h->Pout; bytes = B * nx * nx * 8; break;
        }
        default: return MPCQP_ERR_ARG;
    }
    if (!src) return MPCQP_ERR_ORDER;
    HIPCHK(hipMemcpyAsync(out, src, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return MPCQP_OK;
}

int mpcqp_mhe_nk(mpcqp_mhe h) { return h ? h->Nk : MPCQP_ERR_NULL; }

double mpcqp_mhe_last_ms(mpcqp_mhe h) {
    if (!h || !h->timed) return -1.0;
    DeviceGuard guard_(h->device);
    if (hipEventSynchronize(h->ev1) != hipSuccess) return -1.0;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, h->ev0, h->ev1) != hipSuccess) return -1.0;
    return ms;
}

int mpcqp_mhe_register_columns(mpcqp_mhe h) { return h ? h->d.NX : MPCQP_ERR_NULL; }

}  // extern "C"
