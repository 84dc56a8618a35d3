/* include/mpcqp_mhe.h is valid C and every entry point links (no GPU needed: addresses only). */
#include <stdio.h>
#include "mpcqp_mhe.h"

int main(void) {
    const void* fns[] = {(const void*)mpcqp_mhe_create, (const void*)mpcqp_mhe_destroy, (const void*)mpcqp_mhe_set_model,
                         (const void*)mpcqp_mhe_set_bounds, (const void*)mpcqp_mhe_init, (const void*)mpcqp_mhe_prepare,
                         (const void*)mpcqp_mhe_update, (const void*)mpcqp_mhe_prepare_device,
                         (const void*)mpcqp_mhe_update_device, (const void*)mpcqp_mhe_sync, (const void*)mpcqp_mhe_get,
                         (const void*)mpcqp_mhe_device_ptr, (const void*)mpcqp_mhe_nk, (const void*)mpcqp_mhe_last_ms,
                         (const void*)mpcqp_mhe_register_columns};
    int n = 0;
    for (unsigned i = 0; i < sizeof fns / sizeof fns[0]; ++i) n += fns[i] != 0;
    mpcqp_mhe_dims d = {0};
    mpcqp_mhe h = 0;
    /* argument validation happens before any device call */
    if (mpcqp_mhe_create(0, &h) != MPCQP_ERR_NULL) return 2;
    d.batch = 1; d.nxhat = 17; d.nu = 1; d.nym = 1; d.He = 2;
    (void)d;
    printf("%d mhe entry points\n", n);
    return n == 15 ? 0 : 1;
}
