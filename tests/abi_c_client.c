/* A plain-C client of include/mpcqp.h: proves the header is valid C (no C++ types at the boundary)
 * and that every entry point links against libmpcqp.so.  Built and run by
 * tests/test_abi_and_host.py::test_header_is_plain_c (no GPU needed: it only takes addresses and
 * calls the GPU-free helpers). */
#include <stdio.h>
#include <string.h>
#include "mpcqp.h"

int main(void) {
    const void* fns[] = {
        (const void*)mpcqp_create, (const void*)mpcqp_destroy, (const void*)mpcqp_get_sizes,
        (const void*)mpcqp_set_model, (const void*)mpcqp_set_weights,
        (const void*)mpcqp_set_output_weight_blocks, (const void*)mpcqp_set_custom_constraints,
        (const void*)mpcqp_set_custom_bounds, (const void*)mpcqp_set_flags, (const void*)mpcqp_set_bounds, (const void*)mpcqp_step,
        (const void*)mpcqp_step_device, (const void*)mpcqp_recondense_device, (const void*)mpcqp_get,
        (const void*)mpcqp_last_step_ms, (const void*)mpcqp_last_condense_ms, (const void*)mpcqp_kf_set,
        (const void*)mpcqp_kf_correct, (const void*)mpcqp_kf_predict,
        (const void*)mpcqp_kf_correct_device, (const void*)mpcqp_kf_predict_device,
        (const void*)mpcqp_version, (const void*)mpcqp_strerror, (const void*)mpcqp_last_hip_error};
    unsigned i, n = sizeof fns / sizeof fns[0];
    for (i = 0; i < n; ++i)
        if (!fns[i]) return 1;
    mpcqp_dims d;
    memset(&d, 0, sizeof d);
    if (mpcqp_create(NULL, NULL) != MPCQP_ERR_NULL) return 2;       /* argument check, no device touched */
    if (!mpcqp_strerror(MPCQP_ERR_UNSUPPORTED) || !mpcqp_version()) return 3;
    printf("%s ok (%u entry points, sizeof(mpcqp_dims) = %u)\n", mpcqp_version(), n, (unsigned)sizeof d);
    return 0;
}
