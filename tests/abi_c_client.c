/* A plain-C client of include/mpcqp.h.
 *   abi_c_client                 proves the header is valid C (no C++ types at the boundary) and that
 *                                every entry point links (no GPU needed: it only takes addresses and
 *                                calls the GPU-free helpers);
 *   abi_c_client run <fixture>   drives the whole call sequence a ccall / cgo / JNI host would make --
 *                                create, set_model, set_weights, set_bounds, prepare, step, destroy -- on
 *                                raw float64 fixtures (written by the tests from the golden files) and
 *                                compares the optimum with the fixture's expected one.
 * Built and run by tests/test_abi_and_host.py (against the CPU wave emulator) and by the -m gpu
 * tests (against libmpcqp.so). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mpcqp.h"

static double* rd(FILE* f, size_t n) {
    double* a = (double*)malloc(n * sizeof(double));
    if (!a || fread(a, sizeof(double), n, f) != n) { fprintf(stderr, "short fixture\n"); exit(10); }
    return a;
}

static int run(const char* path) {
    FILE* f = fopen(path, "rb");
    int32_t hd[12];
    if (!f || fread(hd, sizeof(int32_t), 12, f) != 12) return 11;
    const int B = hd[0], nxh = hd[1], nu = hd[2], ny = hd[3], Hp = hd[4], Hc = hd[5], neps = hd[6];
    const int has_u = hd[7], has_du = hd[8], has_ymax = hd[9];
    const size_t nDU = (size_t)nu * Hc, nZ = nDU + neps, nU = (size_t)nu * Hp, nY = (size_t)ny * Hp;
    double *Ahat = rd(f, (size_t)B * nxh * nxh), *Bu = rd(f, (size_t)B * nxh * nu), *C = rd(f, (size_t)B * ny * nxh);
    double *Md = rd(f, B * nY), *Nd = rd(f, B * nDU), *Ld = rd(f, B * nU), *Cw = rd(f, B);
    double *U0min = has_u ? rd(f, B * nU) : NULL, *U0max = has_u ? rd(f, B * nU) : NULL;
    double *DUmin = has_du ? rd(f, B * nDU) : NULL, *DUmax = has_du ? rd(f, B * nDU) : NULL;
    double* Y0max = has_ymax ? rd(f, B * nY) : NULL;
    double *x0 = rd(f, (size_t)B * nxh), *lu = rd(f, (size_t)B * nu), *ry = rd(f, (size_t)B * ny), *Zexp = rd(f, B * nZ);
    fclose(f);

    mpcqp_dims d;
    memset(&d, 0, sizeof d);
    d.batch = B; d.nxhat = nxh; d.nu = nu; d.ny = ny; d.nd = 0; d.Hp = Hp; d.Hc = Hc; d.neps = neps;
    d.flags = MPCQP_FLAG_RY_CONSTANT | MPCQP_FLAG_COLD_START;
    mpcqp_handle h = NULL;
    int rc = mpcqp_create(&d, &h);
    if (rc) { fprintf(stderr, "create: %s %s\n", mpcqp_strerror(rc), mpcqp_last_hip_error()); return 12; }
    mpcqp_sizes sz;
    if (mpcqp_get_sizes(h, &sz) || sz.nZ != (int)nZ || sz.nY != (int)nY) return 13;
    if ((rc = mpcqp_set_model(h, Ahat, Bu, C, NULL, NULL, NULL))) { fprintf(stderr, "set_model %d\n", rc); return 14; }
    if ((rc = mpcqp_set_weights(h, Md, Nd, Ld, neps ? Cw : NULL))) { fprintf(stderr, "set_weights %d\n", rc); return 15; }
    mpcqp_bounds b;
    memset(&b, 0, sizeof b);
    b.U0min = U0min; b.U0max = U0max; b.DUmin = DUmin; b.DUmax = DUmax; b.Y0max = Y0max;
    if ((rc = mpcqp_set_bounds(h, &b))) { fprintf(stderr, "set_bounds %d\n", rc); return 16; }
    uint32_t groups = 0;
    const int kind = mpcqp_prepare(h);
    if (kind < 0 || mpcqp_row_groups(h, &groups) || mpcqp_kernel_kind(h) != kind) return 17;
    double* Z = (double*)calloc(B * nZ, sizeof(double));
    double* u0 = (double*)calloc((size_t)B * nu, sizeof(double));
    int32_t* st = (int32_t*)calloc(B, sizeof(int32_t));
    int32_t* it = (int32_t*)calloc(B, sizeof(int32_t));
    if ((rc = mpcqp_step(h, x0, lu, ry, NULL, NULL, NULL, Z, u0, st, it, NULL))) {
        fprintf(stderr, "step: %s %s\n", mpcqp_strerror(rc), mpcqp_last_hip_error());
        return 18;
    }
    double worst = 0.0;
    for (int p = 0; p < B; ++p) {
        double e = 0.0, s = 1.0;
        if (st[p] != MPCQP_STATUS_OPTIMAL) { fprintf(stderr, "problem %d: status %d\n", p, st[p]); return 19; }
        for (size_t k = 0; k < nDU; ++k) {
            e = fmax(e, fabs(Z[p * nZ + k] - Zexp[p * nZ + k]));
            s = fmax(s, fabs(Zexp[p * nZ + k]));
        }
        for (int c = 0; c < nu; ++c)                          /* getinput!: u0 = dU[0:nu] + lastu0 */
            if (fabs(u0[p * nu + c] - (Z[p * nZ + c] + lu[p * nu + c])) > 1e-12) return 20;
        worst = fmax(worst, e / s);
    }
    printf("run ok: B=%d nZ=%d kernel kind %d row groups 0x%x, mean iterations %.1f, worst rel dU error %.3e\n",
           B, (int)nZ, kind, groups, (double)it[0], worst);
    mpcqp_destroy(h);
    return worst <= 1e-5 ? 0 : 21;
}

int main(int argc, char** argv) {
    const void* fns[] = {
        (const void*)mpcqp_create, (const void*)mpcqp_destroy, (const void*)mpcqp_get_sizes,
        (const void*)mpcqp_set_model, (const void*)mpcqp_set_weights,
        (const void*)mpcqp_set_output_weight_blocks, (const void*)mpcqp_set_custom_constraints,
        (const void*)mpcqp_set_custom_bounds, (const void*)mpcqp_set_flags, (const void*)mpcqp_set_bounds, (const void*)mpcqp_step,
        (const void*)mpcqp_step_device, (const void*)mpcqp_loop_device, (const void*)mpcqp_recondense_device, (const void*)mpcqp_get,
        (const void*)mpcqp_last_step_ms, (const void*)mpcqp_last_condense_ms, (const void*)mpcqp_last_predmat_ms,
        (const void*)mpcqp_kf_set, (const void*)mpcqp_kf_correct, (const void*)mpcqp_kf_predict,
        (const void*)mpcqp_kf_correct_device, (const void*)mpcqp_kf_predict_device,
        (const void*)mpcqp_prepare, (const void*)mpcqp_kernel_kind, (const void*)mpcqp_row_groups,
        (const void*)mpcqp_prebuild, (const void*)mpcqp_last_build_error,
        (const void*)mpcqp_multi_create, (const void*)mpcqp_multi_destroy, (const void*)mpcqp_multi_ndev,
        (const void*)mpcqp_multi_handle, (const void*)mpcqp_multi_shard, (const void*)mpcqp_multi_set_model,
        (const void*)mpcqp_multi_set_weights, (const void*)mpcqp_multi_set_bounds, (const void*)mpcqp_multi_prepare,
        (const void*)mpcqp_multi_step, (const void*)mpcqp_multi_gather_device,
        (const void*)mpcqp_version, (const void*)mpcqp_strerror, (const void*)mpcqp_last_hip_error};
    unsigned i, n = sizeof fns / sizeof fns[0];
    if (argc >= 3 && strcmp(argv[1], "run") == 0) return run(argv[2]);
    for (i = 0; i < n; ++i)
        if (!fns[i]) return 1;
    mpcqp_dims d;
    memset(&d, 0, sizeof d);
    if (mpcqp_create(NULL, NULL) != MPCQP_ERR_NULL) return 2;       /* argument check, no device touched */
    if (mpcqp_prepare(NULL) != MPCQP_ERR_NULL || mpcqp_multi_create(&d, NULL, 1, NULL) != MPCQP_ERR_NULL) return 4;
    if (!mpcqp_strerror(MPCQP_ERR_UNSUPPORTED) || !mpcqp_version()) return 3;
    printf("%s ok (%u entry points, sizeof(mpcqp_dims) = %u)\n", mpcqp_version(), n, (unsigned)sizeof d);
    return 0;
}
