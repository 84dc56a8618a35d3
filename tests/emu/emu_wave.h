// TEST INFRASTRUCTURE ONLY: the CPU "wavefront" of the emulator -- one cooperative fiber per lane (emu_fiber.h), a barrier for
// every wave-level operation.  Shared by emu_launch.cpp and emu_ms.cpp.
#pragma once
#include <cstdlib>
#include <vector>

#include "emu_fiber.h"

#include <math.h>

#include "mpcqp_types.h"

namespace mpcqp {

struct EmuShared {
    LaneFibers& bar = lane_fibers();
    double xd[WAVE];
    int xi[WAVE];
};

struct EmuWave {
    int lane;
    EmuShared* sh;
    void sync() { sh->bar.arrive_and_wait(); }
    void sync_lds() { sync(); }
    double sum(double v) {
        sh->xd[lane] = v; sync();
        double s = 0.0;
        for (int i = 0; i < WAVE; ++i) s += sh->xd[i];
        sync();
        return s;
    }
    double quad_sum(double v) {
        sh->xd[lane] = v; sync();
        const int q = lane & ~3;
        double s = sh->xd[q] + sh->xd[q + 1] + sh->xd[q + 2] + sh->xd[q + 3];
        sync();
        return s;
    }
    double minv(double v) {
        sh->xd[lane] = v; sync();
        double s = sh->xd[0];
        for (int i = 1; i < WAVE; ++i) s = fmin(s, sh->xd[i]);
        sync();
        return s;
    }
    double maxv(double v) {
        sh->xd[lane] = v; sync();
        double s = sh->xd[0];
        for (int i = 1; i < WAVE; ++i) s = fmax(s, sh->xd[i]);
        sync();
        return s;
    }
    int isum(int v) {
        sh->xi[lane] = v; sync();
        int s = 0;
        for (int i = 0; i < WAVE; ++i) s += sh->xi[i];
        sync();
        return s;
    }
    bool any(bool p) { return isum(p ? 1 : 0) != 0; }
    void relane() {}
    double fetch(double v, int src) {
        sh->xd[lane] = v; sync();
        double s = sh->xd[src & (WAVE - 1)];
        sync();
        return s;
    }
    double bcast(double v, int src) {
        sh->xd[lane] = v; sync();
        double s = sh->xd[src];
        sync();
        return s;
    }
};

template <class F>
inline void run_waves(int B, size_t lds_doubles, F body) {
    std::vector<double> smem(lds_doubles + 16, 0.0);
    EmuShared sh;
    int perm[64];
    emu_lane_order(perm);
    sh.bar.run([&](int fiber) {
        EmuWave w{perm[fiber], &sh};
        for (int b = 0; b < B; ++b) {
            body(w, b, smem.data());
            w.sync();
        }
    });
}

}  // namespace mpcqp
