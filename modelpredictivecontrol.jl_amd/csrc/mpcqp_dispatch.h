// mpcqp_dispatch.h -- ahead-of-time specialisations of the step kernel (compile-time dims).
// X(NU, NY, NXH, HP, HC, NEPS, GMASK): any handle whose dimensions, constraint-group pattern
// (Dims::gmask) and default move blocking match one entry runs the specialised kernel; every
// other handle runs the generic runtime-dims kernel (same source, same numerics).
//   C2 = BASELINE.json configs[1]: hard u and Δu box      -> groups box-lo, box-hi, Umin, Umax
//   C3 = BASELINE.json configs[2..3]: hard u, soft ymax    -> groups Umin, Umax, Ymax (the row ϵ >= 0 rides in Ymax: eps_host_group)
#pragma once
#define MPCQP_SPECIALIZATIONS(X)        \
    X(2, 2, 6, 20, 5, 1, 0x0Fu)         \
    X(4, 4, 16, 30, 10, 1, 0x8Cu)
