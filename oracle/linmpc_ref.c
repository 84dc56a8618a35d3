/* CPU ORACLE (test infrastructure, NOT a product path) -- C restatement of the per-step LinMPC
 * path, used (a) as a second, independently written checker of the NumPy oracle at batch sizes
 * NumPy is too slow for, and (b) as the `cpu_baseline` leg of bench.py ("kind": "port").
 *
 * It does what the reference does, the dense way, one controller at a time (OpenMP over
 * controllers):
 *   construction  init_predmat   /root/reference/src/controller/transcription.jl:115-194  (E, K, V)
 *                 init_quadprog  src/controller/construct.jl:837-845                      (H̃)
 *                 init_matconstraint_mpc + relax*  transcription.jl:667-703, construct.jl:999-1083
 *                                (dense A with the slack column, finite rows only)
 *   per step      initpred!      src/controller/execute.jl:247-277                        (F, q̃)
 *                 linconstraint! src/controller/transcription.jl:811-848                  (b)
 *                 optim_objective! -> JuMP/OSQP (execute.jl:466-505) restated as a dense
 *                                float64 Mehrotra predictor-corrector on G = [A; bounds]
 *                 getinput!      src/controller/execute.jl:536-546
 * Scope: nd = 0, diagonal weights, hard u / Δu bounds, output bounds with softness 1 (the
 * configurations of BASELINE.json).  Parity pin: tests/test_oracle_c_port.py checks it against
 * oracle/qp.py (which is pinned on the reference's known-answer tests).
 *
 * Array layout = the C-ABI's: problem-major, column-major inside a problem.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    int B, nxh, nu, ny, Hp, Hc, neps, nZ, nDU, nU, nY, m;   /* m = max rows of G */
    int *jl, *blk;
    double *E;      /* [B][nY*nDU]  row-major */
    double *K;      /* [B][nY*nxh]  row-major */
    double *H;      /* [B][nZ*nZ]   */
    double *Md, *Ld;
    const double *U0min, *U0max, *DUmin, *DUmax, *Y0min, *Y0max;   /* borrowed, may be NULL */
} ref_t;

static void matmul(const double* A, const double* B, double* C, int n, int k, int m) {
    /* C(n x m) = A(n x k) B(k x m), all column-major */
    for (int j = 0; j < m; ++j)
        for (int i = 0; i < n; ++i) {
            double s = 0;
            for (int l = 0; l < k; ++l) s += A[i + n * l] * B[l + k * j];
            C[i + n * j] = s;
        }
}

void* linmpc_ref_create(int B, int nxh, int nu, int ny, int Hp, int Hc, const int* nb, int neps,
                        const double* Ahat, const double* Bu, const double* C, const double* Mdiag,
                        const double* Ndiag, const double* Ldiag, const double* Cwt,
                        const double* U0min, const double* U0max, const double* DUmin,
                        const double* DUmax, const double* Y0min, const double* Y0max) {
    ref_t* r = (ref_t*)calloc(1, sizeof(ref_t));
    r->B = B; r->nxh = nxh; r->nu = nu; r->ny = ny; r->Hp = Hp; r->Hc = Hc; r->neps = neps;
    r->nDU = nu * Hc; r->nZ = r->nDU + neps; r->nU = nu * Hp; r->nY = ny * Hp;
    r->m = 2 * r->nZ + 2 * r->nU + 2 * r->nY;
    r->jl = (int*)calloc(Hc + 1, sizeof(int));
    r->blk = (int*)calloc(Hp, sizeof(int));
    for (int i = 0; i < Hc; ++i) {
        int n = nb ? nb[i] : (i == Hc - 1 ? Hp - Hc + 1 : 1);
        r->jl[i + 1] = r->jl[i] + n;
        for (int t = r->jl[i]; t < r->jl[i + 1]; ++t) r->blk[t] = i;
    }
    const int nY = r->nY, nDU = r->nDU, nZ = r->nZ, nU = r->nU;
    r->E = (double*)calloc((size_t)B * nY * nDU, sizeof(double));
    r->K = (double*)calloc((size_t)B * nY * nxh, sizeof(double));
    r->H = (double*)calloc((size_t)B * nZ * nZ, sizeof(double));
    r->Md = (double*)malloc((size_t)B * nY * sizeof(double));
    r->Ld = (double*)malloc((size_t)B * nU * sizeof(double));
    memcpy(r->Md, Mdiag, (size_t)B * nY * sizeof(double));
    memcpy(r->Ld, Ldiag, (size_t)B * nU * sizeof(double));
    r->U0min = U0min; r->U0max = U0max; r->DUmin = DUmin; r->DUmax = DUmax;
    r->Y0min = Y0min; r->Y0max = Y0max;
#pragma omp parallel for schedule(dynamic, 16)
    for (int b = 0; b < B; ++b) {
        const double* A = Ahat + (size_t)b * nxh * nxh;
        const double* Bm = Bu + (size_t)b * nxh * nu;
        const double* Cm = C + (size_t)b * ny * nxh;
        double* Apow = (double*)malloc((size_t)(Hp + 1) * nxh * nxh * sizeof(double));
        double* Scs = (double*)malloc((size_t)(Hp + 1) * nxh * nxh * sizeof(double));
        double* T1 = (double*)malloc((size_t)nxh * nxh * sizeof(double));
        double* T2 = (double*)malloc((size_t)ny * nxh * sizeof(double));
        /* Apow[j] = A^j, Scs[m] = sum_{l<=m} A^l      transcription.jl:122-128 */
        memset(Apow, 0, (size_t)nxh * nxh * sizeof(double));
        for (int i = 0; i < nxh; ++i) Apow[i + nxh * i] = 1.0;
        memcpy(Scs, Apow, (size_t)nxh * nxh * sizeof(double));
        for (int j = 1; j <= Hp; ++j) {
            matmul(Apow + (size_t)(j - 1) * nxh * nxh, A, Apow + (size_t)j * nxh * nxh, nxh, nxh, nxh);
            for (int i = 0; i < nxh * nxh; ++i)
                Scs[(size_t)j * nxh * nxh + i] = Scs[(size_t)(j - 1) * nxh * nxh + i] + Apow[(size_t)j * nxh * nxh + i];
        }
        double* Eb = r->E + (size_t)b * nY * nDU;
        double* Kb = r->K + (size_t)b * nY * nxh;
        for (int j = 1; j <= Hp; ++j) {           /* K block j = C A^j    :143-147 */
            matmul(Cm, Apow + (size_t)j * nxh * nxh, T2, ny, nxh, nxh);
            for (int a = 0; a < ny; ++a)
                for (int k = 0; k < nxh; ++k) Kb[((j - 1) * ny + a) * nxh + k] = T2[a + ny * k];
        }
        double* Sig = (double*)malloc((size_t)Hp * ny * nu * sizeof(double));   /* C S(m) Bu */
        for (int mm = 0; mm < Hp; ++mm) {
            matmul(Cm, Scs + (size_t)mm * nxh * nxh, T2, ny, nxh, nxh);
            for (int a = 0; a < ny; ++a)
                for (int c = 0; c < nu; ++c) {
                    double s = 0;
                    for (int l = 0; l < nxh; ++l) s += T2[a + ny * l] * Bm[l + nxh * c];
                    Sig[(mm * ny + a) * nu + c] = s;
                }
        }
        for (int j = 0; j < Hc; ++j)              /* E   :153-165 */
            for (int t = r->jl[j]; t < Hp; ++t)
                for (int a = 0; a < ny; ++a)
                    for (int c = 0; c < nu; ++c)
                        Eb[(t * ny + a) * nDU + j * nu + c] = Sig[((t - r->jl[j]) * ny + a) * nu + c];
        /* H̃ = 2(E'ME + N + Pu'L Pu) (+) 2C          construct.jl:842 */
        double* Hb = r->H + (size_t)b * nZ * nZ;
        const double* Md = Mdiag + (size_t)b * nY;
        const double* Ld = Ldiag + (size_t)b * nU;
        for (int i = 0; i < nDU; ++i)
            for (int k = 0; k <= i; ++k) {
                double s = 0;
                for (int rr = 0; rr < nY; ++rr) s += Eb[rr * nDU + i] * Md[rr] * Eb[rr * nDU + k];
                int ji = i / nu, ci = i % nu, jk = k / nu, ck = k % nu;
                if (ci == ck)
                    for (int t = r->jl[ji > jk ? ji : jk]; t < Hp; ++t) s += Ld[t * nu + ci];
                if (i == k) s += Ndiag[(size_t)b * nDU + i];
                Hb[i * nZ + k] = Hb[k * nZ + i] = 2.0 * s;
            }
        if (neps) Hb[(nZ - 1) * nZ + nZ - 1] = 2.0 * Cwt[b];
        free(Apow); free(Scs); free(T1); free(T2); free(Sig);
    }
    return r;
}

void linmpc_ref_destroy(void* p) {
    ref_t* r = (ref_t*)p;
    if (!r) return;
    free(r->jl); free(r->blk); free(r->E); free(r->K); free(r->H); free(r->Md); free(r->Ld);
    free(r);
}

/* Active-set polish of the interior-point iterate (same rule as Step::polish of the kernels):
 * once mu <= POL_MU the rows with lam_i > s_i are taken as the active set A and the
 * equality-constrained QP on A is solved by Newton steps on its augmented Lagrangian,
 *     (H + rho G_A'G_A) dz = -(H z + q + G_A'lam) - rho G_A'(G_A z - h_A),   lam += rho (G_A (z + dz) - h_A),
 * with exactly evaluated residuals each round; the point is accepted when it satisfies the KKT
 * conditions of the inequality-constrained QP (multipliers >= 0 on A, inactive rows feasible). */
static double POL_MU = 1e-7, POL_RHO = 1e10, POL_RD = 1e-14, POL_RP = 1e-13, POL_LAM = 1e-12, POL_SL = 1e-11;
static int POL_ROUNDS = 8;
static double TERM_PFAC = 10.0, TERM_PSTALL = 1e-9;   /* primal residual target (x res_tol) and stall ceiling (x nh) */
void linmpc_ref_term_params(double pfac, double pstall) { TERM_PFAC = pfac; TERM_PSTALL = pstall; }
void linmpc_ref_polish_params(double mu, double rho, int rounds) { POL_MU = mu; POL_RHO = rho; POL_ROUNDS = rounds; }
void linmpc_ref_polish_tols(double rd, double rp, double lam, double sl) { POL_RD = rd; POL_RP = rp; POL_LAM = lam; POL_SL = sl; }

static int chol(double* A, int n) {     /* in place, lower, row-major full */
    for (int k = 0; k < n; ++k) {
        double v = A[k * n + k];
        const double thr = 1e-14 * fabs(v);   /* same pivot threshold as the kernels (Step::cholesky) */
        for (int j = 0; j < k; ++j) v -= A[k * n + j] * A[k * n + j];
        if (!(v > thr)) return -1;
        double d = sqrt(v);
        A[k * n + k] = d;
        for (int i = k + 1; i < n; ++i) {
            double s = A[i * n + k];
            for (int j = 0; j < k; ++j) s -= A[i * n + j] * A[k * n + j];
            A[i * n + k] = s / d;
        }
    }
    return 0;
}

static void chol_solve(const double* L, double* x, int n) {
    for (int i = 0; i < n; ++i) {
        double s = x[i];
        for (int j = 0; j < i; ++j) s -= L[i * n + j] * x[j];
        x[i] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = x[i];
        for (int j = i + 1; j < n; ++j) s -= L[j * n + i] * x[j];
        x[i] = s / L[i * n + i];
    }
}

/* One control period for controllers [0, B).  Ry: (ny,B) held over Hp.  Z: in = previous optimum,
 * out = optimum (or shifted warm start on status 2).  Returns the number of status != 0. */
int linmpc_ref_step(void* p, const double* xhat0, const double* lastu0, const double* Ry,
                    double* Z, double* u0, int* status, int* iters, int cold, int nthreads,
                    double gap_tol, double res_tol, double delta0, int max_iter) {
    ref_t* r = (ref_t*)p;
    const int B = r->B, nxh = r->nxh, nu = r->nu, ny = r->ny, Hp = r->Hp, Hc = r->Hc;
    const int nZ = r->nZ, nDU = r->nDU, nU = r->nU, nY = r->nY, neps = r->neps, mmax = r->m;
    int nbad = 0;
    const double lam0 = 10.0;      /* starting point: s = max(h - G z, 1), lam = lam0 / s */
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel reduction(+ : nbad)
    {
        double* G = (double*)malloc((size_t)mmax * nZ * sizeof(double));
        double* h = (double*)malloc(mmax * sizeof(double));
        double* s = (double*)malloc(mmax * sizeof(double));
        double* lam = (double*)malloc(mmax * sizeof(double));
        double* rp = (double*)malloc(mmax * sizeof(double));
        double* gd = (double*)malloc(mmax * sizeof(double));
        double* pp = (double*)malloc(mmax * sizeof(double));
        double* Dt = (double*)malloc(mmax * sizeof(double));
        double* wv = (double*)malloc(mmax * sizeof(double));
        double* dsv = (double*)malloc(mmax * sizeof(double));
        double* dlv = (double*)malloc(mmax * sizeof(double));
        double* F = (double*)malloc(nY * sizeof(double));
        double* Phi = (double*)malloc((size_t)nZ * nZ * sizeof(double));
        double *q = (double*)malloc(nZ * sizeof(double)), *z = (double*)malloc(nZ * sizeof(double));
        double *zs = (double*)malloc(nZ * sizeof(double)), *dz = (double*)malloc(nZ * sizeof(double));
        double *rd = (double*)malloc(nZ * sizeof(double)), *gt = (double*)malloc(nZ * sizeof(double));
        double *zp = (double*)malloc(nZ * sizeof(double)), *lp = (double*)malloc(mmax * sizeof(double));
        double *rpa = (double*)malloc(mmax * sizeof(double));
        int* act = (int*)malloc(mmax * sizeof(int));
#pragma omp for schedule(dynamic, 8)
        for (int b = 0; b < B; ++b) {
            double delta = delta0;       /* may be raised for this controller, see the factorisation */
            const double* Eb = r->E + (size_t)b * nY * nDU;
            const double* Kb = r->K + (size_t)b * nY * nxh;
            const double* Hb = r->H + (size_t)b * nZ * nZ;
            const double* x0 = xhat0 + (size_t)b * nxh;
            const double* lu = lastu0 + (size_t)b * nu;
            /* initpred!: F = K x̂0 + V lastu0 ; q̃ = 2[(M E)'(F - R̂y) + (L Pu)'(Tu lastu0)] */
            for (int rr = 0; rr < nY; ++rr) {
                double sacc = 0;
                for (int k = 0; k < nxh; ++k) sacc += Kb[rr * nxh + k] * x0[k];
                for (int c = 0; c < nu; ++c) sacc += Eb[rr * nDU + c] * lu[c];   /* V = E[:,1:nu] */
                F[rr] = sacc;
            }
            for (int k = 0; k < nZ; ++k) q[k] = 0;
            for (int rr = 0; rr < nY; ++rr) {
                double cy = r->Md[(size_t)b * nY + rr] * (F[rr] - Ry[(size_t)b * ny + rr % ny]);
                for (int k = 0; k < nDU; ++k) q[k] += Eb[rr * nDU + k] * cy;
            }
            for (int k = 0; k < nDU; ++k) {
                int j = k / nu, c = k % nu;
                double sacc = 0;
                for (int t = r->jl[j]; t < Hp; ++t) sacc += r->Ld[(size_t)b * nU + t * nu + c] * lu[c];
                q[k] = 2.0 * (q[k] + sacc);
            }
            /* linconstraint! + i_b: dense G, h of the finite rows */
            int m = 0;
#define ROW_BEGIN() double* g = G + (size_t)m * nZ; memset(g, 0, nZ * sizeof(double))
            for (int k = 0; k < nZ; ++k) {           /* variable bounds */
                double lo = -INFINITY, hi = INFINITY;
                if (k < nDU) {
                    if (r->DUmin) lo = r->DUmin[(size_t)b * nDU + k];
                    if (r->DUmax) hi = r->DUmax[(size_t)b * nDU + k];
                } else lo = 0.0;
                if (isfinite(lo)) { ROW_BEGIN(); g[k] = -1; h[m++] = -lo; }
                if (isfinite(hi)) { ROW_BEGIN(); g[k] = 1; h[m++] = hi; }
            }
            for (int sgn = -1; sgn <= 1; sgn += 2) {  /* U rows: Pu = held cumulative sum */
                const double* bd = sgn < 0 ? r->U0min : r->U0max;
                if (!bd) continue;
                for (int rr = 0; rr < nU; ++rr) {
                    double v = bd[(size_t)b * nU + rr];
                    if (!isfinite(v)) continue;
                    int t = rr / nu, c = rr % nu;
                    ROW_BEGIN();
                    for (int j = 0; j <= r->blk[t]; ++j) g[j * nu + c] = sgn;
                    h[m++] = sgn < 0 ? -v + lu[c] : v - lu[c];
                }
            }
            for (int sgn = -1; sgn <= 1; sgn += 2) {  /* Ŷ rows, softness 1 */
                const double* bd = sgn < 0 ? r->Y0min : r->Y0max;
                if (!bd) continue;
                for (int rr = 0; rr < nY; ++rr) {
                    double v = bd[(size_t)b * nY + rr];
                    if (!isfinite(v)) continue;
                    ROW_BEGIN();
                    for (int k = 0; k < nDU; ++k) g[k] = sgn * Eb[rr * nDU + k];
                    if (neps) g[nZ - 1] = -1.0;
                    h[m++] = sgn < 0 ? -v + F[rr] : v - F[rr];
                }
            }
            /* warm start */
            double* Zb = Z + (size_t)b * nZ;
            for (int k = 0; k < nZ; ++k) {
                double v = 0;
                if (!cold) { if (k < nDU - nu) v = Zb[k + nu]; else if (k >= nDU) v = Zb[k]; }
                zs[k] = z[k] = v;
            }
            int st = 1, it = 0, npol = 0;
            double polmu_next = POL_MU;
            double laststep = 1e300;     /* alpha |dU|_inf / max(1, |dU|_inf) of the previous iteration */
            double rdn_prev = 1e300, rpn_prev = 1e300, lastscale = 1.0;   /* previous dual residual, 1 - alpha of the previous step */
            double nh = 1.0, rpn = 0;
            for (int i = 0; i < m; ++i) {       /* starting point: s = max(h - G z, 1), lam = 10 / s */
                double a = 0;
                for (int k = 0; k < nZ; ++k) a += G[(size_t)i * nZ + k] * z[k];
                s[i] = fmax(h[i] - a, 1.0);
                lam[i] = lam0 / s[i];
                if (fabs(h[i]) + 1 > nh) nh = fabs(h[i]) + 1;
            }
            if (m == 0) {
                memcpy(Phi, Hb, (size_t)nZ * nZ * sizeof(double));
                chol(Phi, nZ);
                for (int k = 0; k < nZ; ++k) z[k] = -q[k];
                chol_solve(Phi, z, nZ);
                st = 0;
            }
            for (int pass = 0; m > 0 && pass < max_iter; ++pass) {
                /* residuals */
                double mu = 0, rdn = 0, ndd = 0;
                rpn = 0;
                for (int i = 0; i < m; ++i) {
                    double a = 0;
                    for (int k = 0; k < nZ; ++k) a += G[(size_t)i * nZ + k] * z[k];
                    rp[i] = a + s[i] - h[i];
                    if (fabs(rp[i]) > rpn) rpn = fabs(rp[i]);
                    mu += s[i] * lam[i];
                }
                mu /= m;
                for (int k = 0; k < nZ; ++k) {
                    double hz = 0, gl = 0;
                    for (int j = 0; j < nZ; ++j) hz += Hb[k * nZ + j] * z[j];
                    for (int i = 0; i < m; ++i) gl += G[(size_t)i * nZ + k] * lam[i];
                    rd[k] = hz + q[k] + gl;
                    if (fabs(rd[k]) > rdn) rdn = fabs(rd[k]);
                    double sc = fmax(fabs(q[k]), fmax(fabs(hz), fabs(gl)));
                    if (sc > ndd) ndd = sc;
                }
                ndd += 1.0;
                {
                    it = pass;
                    if (!(mu == mu)) { st = 2; break; }
                    /* converged: gap, residuals, and a last Newton step that no longer moves the inputs */
                    /* (a dual residual stuck at the float64 floor of the normal equations although the last
                       step was nearly full counts as converged; the step criterion vouches for z) */
                    int stalled = rdn >= 0.5 * rdn_prev && lastscale <= 0.1;
                    rdn_prev = rdn;
                    int pstalled = rpn >= 0.5 * rpn_prev && lastscale <= 0.1 && rpn <= TERM_PSTALL * nh;
                    rpn_prev = rpn;
                    if (getenv("LINMPC_REF_DEBUG") && b == atoi(getenv("LINMPC_REF_DEBUG")))
                        fprintf(stderr, "  [ref %d] it %3d mu %.3e rd %.3e (rel %.3e%s) rp %.3e (rel %.3e%s) laststep %.3e lastscale %.3e delta %.1e\n", b, pass, mu,
                                rdn, rdn / ndd, stalled ? " stalled" : "", rpn, rpn / nh, pstalled ? " stalled" : "", laststep, lastscale, delta);
                    if (mu <= gap_tol && (rdn <= res_tol * ndd || stalled) && (rpn <= TERM_PFAC * res_tol * nh || pstalled) && laststep <= 1e-6) { st = 0; break; }
                }
                if (POL_MU > 0 && mu <= polmu_next && rpn <= 1e-6 * nh && npol < 4) {
                    polmu_next = 1e-2 * mu;
                    ++npol;
                    for (int i = 0; i < m; ++i) { act[i] = lam[i] > s[i]; lp[i] = act[i] ? lam[i] : 0.0; }
                    memcpy(Phi, Hb, (size_t)nZ * nZ * sizeof(double));
                    for (int i = 0; i < m; ++i) {
                        if (!act[i]) continue;
                        const double* g = G + (size_t)i * nZ;
                        for (int k = 0; k < nZ; ++k) {
                            double gk = POL_RHO * g[k];
                            if (gk == 0) continue;
                            for (int j = 0; j <= k; ++j) Phi[k * nZ + j] += gk * g[j];
                        }
                    }
                    if (chol(Phi, nZ) == 0) {
                        memcpy(zp, z, nZ * sizeof(double));
                        int ok = 0;
                        for (int round = 0; round <= POL_ROUNDS; ++round) {
                            double rpan = 0, rdn2 = 0, ndd2 = 0;
                            for (int i = 0; i < m; ++i) {
                                if (!act[i]) continue;
                                double a = 0;
                                const double* g = G + (size_t)i * nZ;
                                for (int k = 0; k < nZ; ++k) a += g[k] * zp[k];
                                rpa[i] = a - h[i];
                                if (fabs(rpa[i]) > rpan) rpan = fabs(rpa[i]);
                            }
                            for (int k = 0; k < nZ; ++k) {
                                double hz = 0, gl = 0;
                                for (int j = 0; j < nZ; ++j) hz += Hb[k * nZ + j] * zp[j];
                                for (int i = 0; i < m; ++i) if (act[i]) gl += G[(size_t)i * nZ + k] * lp[i];
                                gt[k] = hz + q[k] + gl;
                                if (fabs(gt[k]) > rdn2) rdn2 = fabs(gt[k]);
                                double sc = fmax(fabs(q[k]), fmax(fabs(hz), fabs(gl)));
                                if (sc > ndd2) ndd2 = sc;
                            }
                            if (rpan <= POL_RP * nh && rdn2 <= POL_RD * (1.0 + ndd2)) { ok = 1; break; }
                            if (round == POL_ROUNDS) break;
                            for (int k = 0; k < nZ; ++k) dz[k] = -gt[k];
                            for (int i = 0; i < m; ++i) {
                                if (!act[i]) continue;
                                const double* g = G + (size_t)i * nZ;
                                double c = POL_RHO * rpa[i];
                                for (int k = 0; k < nZ; ++k) dz[k] -= g[k] * c;
                            }
                            chol_solve(Phi, dz, nZ);
                            for (int k = 0; k < nZ; ++k) zp[k] += dz[k];
                            for (int i = 0; i < m; ++i) {
                                if (!act[i]) continue;
                                double a = 0;
                                const double* g = G + (size_t)i * nZ;
                                for (int k = 0; k < nZ; ++k) a += g[k] * dz[k];
                                lp[i] += POL_RHO * (rpa[i] + a);
                            }
                        }
                        if (ok) {       /* KKT conditions of the inequality-constrained QP */
                            double lmax = 0;
                            for (int i = 0; i < m; ++i) if (act[i] && fabs(lp[i]) > lmax) lmax = fabs(lp[i]);
                            for (int i = 0; i < m && ok; ++i) {
                                if (act[i]) { if (lp[i] < -POL_LAM * (1.0 + lmax)) ok = 0; }
                                else {
                                    double a = 0;
                                    const double* g = G + (size_t)i * nZ;
                                    for (int k = 0; k < nZ; ++k) a += g[k] * zp[k];
                                    if (h[i] - a < -POL_SL * nh) ok = 0;
                                }
                            }
                        }
                        if (ok) { memcpy(z, zp, nZ * sizeof(double)); st = 0; it = pass + npol; break; }
                    }
                }
                /* a pivot below its threshold: Phi left float64's range; redo the factorisation with a
                   100x larger dual regularisation (same rule as Step::run of the kernels) */
                int broke = 0;
                for (int attempt = 0;; ++attempt) {
                    /* Phi = H + G' D~ G */
                    for (int i = 0; i < m; ++i) {
                        double D = lam[i] / s[i];
                        wv[i] = 1.0 / (1.0 + delta * D);
                        Dt[i] = D * wv[i];
                    }
                    memcpy(Phi, Hb, (size_t)nZ * nZ * sizeof(double));
                    for (int i = 0; i < m; ++i) {
                        const double* g = G + (size_t)i * nZ;
                        for (int k = 0; k < nZ; ++k) {
                            double gk = Dt[i] * g[k];
                            if (gk == 0) continue;
                            for (int j = 0; j <= k; ++j) Phi[k * nZ + j] += gk * g[j];
                        }
                    }
                    broke = chol(Phi, nZ);
                    if (!broke || attempt == 2 || delta >= 1e-8) break;
                    delta *= 100.0;
                }
                if (broke) { st = 2; break; }
                double smu = 0;
                for (int phase = 0; phase < 2; ++phase) {
                    /* rhs = -rd + G'(w rc/s - D~ rp) */
                    for (int k = 0; k < nZ; ++k) gt[k] = -rd[k];
                    for (int i = 0; i < m; ++i) {
                        double rc = s[i] * lam[i] + (phase ? pp[i] - smu : 0.0);
                        double c = wv[i] * rc / s[i] - Dt[i] * rp[i];
                        const double* g = G + (size_t)i * nZ;
                        for (int k = 0; k < nZ; ++k) gt[k] += g[k] * c;
                    }
                    memcpy(dz, gt, nZ * sizeof(double));
                    chol_solve(Phi, dz, nZ);
                    double amin = 1e300;
                    for (int i = 0; i < m; ++i) {
                        double a = 0;
                        const double* g = G + (size_t)i * nZ;
                        for (int k = 0; k < nZ; ++k) a += g[k] * dz[k];
                        gd[i] = a;
                    }
                    for (int i = 0; i < m; ++i) {
                        double rc = s[i] * lam[i] + (phase ? pp[i] - smu : 0.0);
                        double dl = -wv[i] * rc / s[i] + Dt[i] * (rp[i] + gd[i]);
                        double ds = -wv[i] * ((rp[i] + gd[i]) + delta * rc / s[i]);   /* = -(rc + s dl)/lam */
                        if (ds < 0) amin = fmin(amin, -s[i] / ds);
                        if (dl < 0) amin = fmin(amin, -lam[i] / dl);
                        dsv[i] = ds;
                        dlv[i] = dl;
                    }
                    if (!phase) {
                        double aaff = fmin(1.0, amin), mas = 0;
                        for (int i = 0; i < m; ++i) {
                            mas += (s[i] + aaff * dsv[i]) * (lam[i] + aaff * dlv[i]);
                            pp[i] = dsv[i] * dlv[i];
                        }
                        double sig = (mas / m) / mu;
                        smu = sig * sig * sig * mu;
                    } else {
                        /* fraction to the boundary: 0.9999 if the iterate it leads to stays in the
                           wide neighbourhood min s_i lam_i >= 0.01 mu, else 0.99 */
                        double alpha = fmin(1.0, 0.9999 * amin), pmin = 1e300, psum = 0;
                        for (int i = 0; i < m; ++i) {
                            double p = (s[i] + alpha * dsv[i]) * (lam[i] + alpha * dlv[i]);
                            psum += p;
                            if (p < pmin) pmin = p;
                        }
                        if (!(pmin * m >= 0.01 * psum)) alpha = fmin(1.0, 0.99 * amin);
                        for (int i = 0; i < m; ++i) { s[i] += alpha * dsv[i]; lam[i] += alpha * dlv[i]; }
                        {
                            double zm = 1.0, dm = 0.0;
                            for (int k = 0; k < nDU; ++k) { zm = fmax(zm, fabs(z[k])); dm = fmax(dm, fabs(alpha * dz[k])); }
                            /* a step cut short by the boundary (alpha < 1/2) says nothing about convergence: only a (nearly) full
                               Newton step that no longer moves the inputs does.  (Instance 99 of shape 8,2,2,60,40: the iteration
                               crept along at |dU step| = 6e-6 for 40 iterations, then a blocked step, alpha -> 0, passed the
                               test 4.9e-4 from the optimum -- adjudicated in 60-digit arithmetic, tests/golden/hp_optima.json) */
                            laststep = alpha >= 0.5 ? dm / zm : 1e300;
                            lastscale = 1.0 - alpha;
                        }
                        for (int k = 0; k < nZ; ++k) z[k] += alpha * dz[k];
                    }
                }
            }
            if (m > 0 && st == 1 && !(rpn <= 1e-6 * nh)) st = 2;
            if (st == 2) memcpy(z, zs, nZ * sizeof(double));
            memcpy(Zb, z, nZ * sizeof(double));
            for (int c = 0; c < nu; ++c) u0[(size_t)b * nu + c] = z[c] + lu[c];
            status[b] = st;
            if (iters) iters[b] = it;
            if (st) ++nbad;
        }
        free(G); free(h); free(s); free(lam); free(rp); free(gd); free(pp); free(Dt); free(wv); free(dsv); free(dlv);
        free(F); free(Phi); free(q); free(z); free(zs); free(dz); free(rd); free(gt); free(zp); free(lp); free(rpa); free(act);
    }
    return nbad;
}

int linmpc_ref_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
