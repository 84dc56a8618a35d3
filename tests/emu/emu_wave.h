// TEST INFRASTRUCTURE ONLY: the CPU "wavefront" of the emulator -- one cooperative fiber per lane (emu_fiber.h), a barrier for
// every wave-level operation.  Shared by emu_launch.cpp and emu_ms.cpp.
#pragma once
#include <cstdio>
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
    unsigned tg[2][WAVE], cn[2][WAVE];     // kind and index of the wave-level operation every lane is in (EmuWave::enter)
};

struct EmuWave {
    static constexpr int NTEAM = 1, WV = 0;      // (teams of wavefronts are a device-only form: mpcqp_devwave.h)
    void post(int, int = 0, int = 0, int = 0, int = 0, double = 0.0) {}
    void join() {}
    int lane;
    EmuShared* sh;
    unsigned n = 0;
    // Every wave-level operation must be reached by ALL lanes in the same order (on the GPU a lane that skips one reads a
    // stale DPP / readlane value; here the barriers would pair up operations that do not belong together and the test
    // would fail somewhere else, or crash: ADVICE r4).  Each lane posts (kind, index) of the operation it enters and,
    // past the operation's barrier, checks that every lane posted the same -- a divergent call site aborts with a message.
    void enter(unsigned kind) { ++n; sh->tg[n & 1][lane] = kind; sh->cn[n & 1][lane] = n; }
    void check(const char* what) const {
        for (int i = 0; i < WAVE; ++i)
            if (sh->tg[n & 1][i] != sh->tg[n & 1][lane] || sh->cn[n & 1][i] != n) {
                fprintf(stderr, "[emu] lanes disagree on the sequence of wave-level operations: lane %d in %s (operation %u, kind %u), "
                        "lane %d in operation %u of kind %u\n", lane, what, n, sh->tg[n & 1][lane], i, sh->cn[n & 1][i], sh->tg[n & 1][i]);
                fflush(stderr);
                abort();
            }
    }
    void bar() { sh->bar.arrive_and_wait(); }
    void sync() { enter(0); bar(); check("sync"); }
    void sync_lds() { sync(); }
    double sum(double v) {
        enter(1); sh->xd[lane] = v; bar(); check("sum");
        double s = 0.0;
        for (int i = 0; i < WAVE; ++i) s += sh->xd[i];
        bar();
        return s;
    }
    double quad_sum(double v) {
        enter(2); sh->xd[lane] = v; bar(); check("quad_sum");
        const int q = lane & ~3;
        double s = sh->xd[q] + sh->xd[q + 1] + sh->xd[q + 2] + sh->xd[q + 3];
        bar();
        return s;
    }
    double minv(double v) {
        enter(3); sh->xd[lane] = v; bar(); check("minv");
        double s = sh->xd[0];
        for (int i = 1; i < WAVE; ++i) s = fmin(s, sh->xd[i]);
        bar();
        return s;
    }
    double maxv(double v) {
        enter(4); sh->xd[lane] = v; bar(); check("maxv");
        double s = sh->xd[0];
        for (int i = 1; i < WAVE; ++i) s = fmax(s, sh->xd[i]);
        bar();
        return s;
    }
    int isum(int v) {
        enter(5); sh->xi[lane] = v; bar(); check("isum");
        int s = 0;
        for (int i = 0; i < WAVE; ++i) s += sh->xi[i];
        bar();
        return s;
    }
    bool any(bool p) { return isum(p ? 1 : 0) != 0; }
    void relane() {}
    double fetch(double v, int src) {
        enter(6); sh->xd[lane] = v; bar(); check("fetch");
        double s = sh->xd[src & (WAVE - 1)];
        bar();
        return s;
    }
    double bcast(double v, int src) {
        enter(7); sh->xd[lane] = v; bar(); check("bcast");
        double s = sh->xd[src];
        bar();
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
