// TEST INFRASTRUCTURE ONLY: runs the MovingHorizonEstimator kernel bodies (csrc/mhe_bodies.h) on the
// CPU, one cooperative fiber per lane of a 64-wide "wavefront" (emu_fiber.h), wavefronts one after the other.
#include <algorithm>
#include <atomic>
#include <barrier>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "emu_fiber.h"
#include "mhe_bodies.h"
#include "mhe_launch.h"
#include "mpcqp_launch.h"
#include "mpcqp_small_bodies.h"

namespace mpcqp {
namespace mhe {

struct EmuShared {
    LaneFibers& bar = lane_fibers();
    double xd[2][WAVE];
    unsigned cn[2][WAVE];                // index of the cross-lane operation every lane is in
    unsigned long calls[WAVE] = {};      // cross-lane operations of every lane (MPCQP_EMU_WATCHDOG)
};

// every cross-lane operation writes buffer (n % 2) of its n-th call, waits once, reads: a lane can
// only reach call n + 2 (same buffer) after all lanes passed the barrier of call n + 1, i.e. after
// all of them finished reading call n
struct EmuWave {
    int lane;
    EmuShared* sh;
    unsigned n = 0;
    void sync() { sh->bar.arrive_and_wait(); }
    double* xchg(double v) {
        sh->cn[n & 1][lane] = n;
        double* buf = sh->xd[n++ & 1];
        buf[lane] = v;
        ++sh->calls[lane];
        sh->bar.arrive_and_wait();
        // every lane must be in the SAME cross-lane operation (a product under a per-estimator condition once was not:
        // mhe_bodies.h write_outputs): a divergent call site aborts here with a message instead of corrupting
        // a fiber stack somewhere later
        for (int i = 0; i < WAVE; ++i)
            if (sh->cn[(n - 1) & 1][i] != n - 1) {
                fprintf(stderr, "[emu] lanes disagree on the sequence of cross-lane operations: lane %d in operation %u, lane %d in %u\n",
                        lane, n - 1, i, sh->cn[(n - 1) & 1][i]);
                fflush(stderr);
                abort();
            }
        return buf;
    }
    template <int C>
    double rowbc(double v) { return xchg(v)[(lane & ~(RL - 1)) + C]; }
    template <class T>
    T* uniform(T* p) const { return p; }
    struct Buf { double* p; size_t bytes; };
    static constexpr unsigned BUF_OOB = 0xFFFFFFF0u;
    Buf make_buf(double* base, size_t bytes) const { return Buf{base, bytes}; }
    double bload(Buf b, unsigned voff, int soff) const {
        const size_t o = (size_t)voff + (size_t)soff;
        return (voff == BUF_OOB || o + 8 > b.bytes) ? 0.0 : *(const double*)((const char*)b.p + o);
    }
    void bstore(Buf b, unsigned voff, int soff, double v) const {
        const size_t o = (size_t)voff + (size_t)soff;
        if (voff != BUF_OOB && o + 8 <= b.bytes) *(double*)((char*)b.p + o) = v;
    }
    template <int L0, int L1, int L2, int L3>
    void fmabc4(double& acc, double x0, double x1, double x2, double x3, double y0, double y1, double y2, double y3) {
        acc = fma(rowbc<L0>(x0), y0, acc); acc = fma(rowbc<L1>(x1), y1, acc);
        acc = fma(rowbc<L2>(x2), y2, acc); acc = fma(rowbc<L3>(x3), y3, acc);
    }
    template <int L0, int L1, int L2, int L3>
    void rank1bc4(double& a0, double& a1, double& a2, double& a3, double x, double y0, double y1, double y2, double y3) {
        const double b0 = rowbc<L0>(x), b1 = rowbc<L1>(x), b2 = rowbc<L2>(x), b3 = rowbc<L3>(x);
        a0 = fma(b0, y0, a0); a1 = fma(b1, y1, a1); a2 = fma(b2, y2, a2); a3 = fma(b3, y3, a3);
    }
    template <int K>
    void gjrow4(double& a0, double& a1, double& a2, double& a3, double m, double g) {
        const double b0 = rowbc<K>(a0), b1 = rowbc<K>(a1), b2 = rowbc<K>(a2), b3 = rowbc<K>(a3);
        a0 = fma(b0, g, a0 * m); a1 = fma(b1, g, a1 * m); a2 = fma(b2, g, a2 * m); a3 = fma(b3, g, a3 * m);
    }
    template <int L0, int L1, int L2, int L3>
    void fmsbc4(double& acc, double x0, double x1, double x2, double x3, double y0, double y1, double y2, double y3) {
        acc = fma(rowbc<L0>(x0), -y0, acc); acc = fma(rowbc<L1>(x1), -y1, acc);
        acc = fma(rowbc<L2>(x2), -y2, acc); acc = fma(rowbc<L3>(x3), -y3, acc);
    }
    template <int K>
    void gjacc4(double& a0, double& a1, double& a2, double& a3, double g) {
        const double b0 = rowbc<K>(a0), b1 = rowbc<K>(a1), b2 = rowbc<K>(a2), b3 = rowbc<K>(a3);
        a0 = fma(b0, g, a0); a1 = fma(b1, g, a1); a2 = fma(b2, g, a2); a3 = fma(b3, g, a3);
    }
    template <class Op>
    double rowred(double v, Op op) {
        const double* buf = xchg(v);
        const int r0 = lane & ~(RL - 1);
        double s = buf[r0];
        for (int i = 1; i < RL; ++i) s = op(s, buf[r0 + i]);
        return s;
    }
    double rsum(double v) { return rowred(v, [](double x, double y) { return x + y; }); }
    double rmin(double v) { return rowred(v, [](double x, double y) { return fmin(x, y); }); }
    double rmax(double v) { return rowred(v, [](double x, double y) { return fmax(x, y); }); }
    bool any(bool p) {
        const double* buf = xchg(p ? 1.0 : 0.0);
        for (int i = 0; i < WAVE; ++i) if (buf[i] != 0.0) return true;
        return false;
    }
};

template <class F>
static void run_waves(int nwaves, size_t lds_doubles, F body) {
    std::vector<double> smem(lds_doubles + 16, 0.0);
    EmuShared sh;
    std::atomic<bool> stop{false};
    std::thread dog;
    if (getenv("MPCQP_EMU_WATCHDOG"))      // lanes that stopped agreeing on the number of cross-lane operations
        dog = std::thread([&] {
            unsigned long last = 0;
            while (!stop) {
                std::this_thread::sleep_for(std::chrono::seconds(3));
                unsigned long mn = ~0ul, mx = 0;
                for (int i = 0; i < WAVE; ++i) { mn = std::min(mn, sh.calls[i]); mx = std::max(mx, sh.calls[i]); }
                if (mx == last && mx != mn) {
                    fprintf(stderr, "[emu watchdog] lanes disagree:");
                    for (int i = 0; i < WAVE; ++i) fprintf(stderr, " %lu", sh.calls[i]);
                    fprintf(stderr, "\n");
                }
                if (getenv("MPCQP_EMU_WATCHDOG")[0] == '2') fprintf(stderr, "[emu watchdog] %lu..%lu cross-lane ops\n", mn, mx);
                last = mx;
            }
        });
    int perm[64];
    emu_lane_order(perm);
    sh.bar.run([&](int fiber) {
        EmuWave w{perm[fiber], &sh};
        for (int wv = 0; wv < nwaves; ++wv) {
            body(w, wv, smem.data());
            w.sync();
        }
    });
    stop = true;
    if (dog.joinable()) dog.join();
}

#define MHE_DISPATCH(NXV, CALL)                          \
    switch (NXV) {                                       \
        case 4: { constexpr int NX = 4; CALL; } break;   \
        case 8: { constexpr int NX = 8; CALL; } break;   \
        case 12: { constexpr int NX = 12; CALL; } break; \
        case 16: { constexpr int NX = 16; CALL; } break; \
        default: return hipErrorInvalidValue;            \
    }

hipError_t launch_setup(const Dims& d, const Raw& in, double* cst, hipStream_t) {
    MHE_DISPATCH(d.NX, run_waves(d.nwaves, 0, [&](EmuWave& w, int wv, double*) { setup_body<EmuWave, NX>(w, d, in, cst, wv); }));
    return hipSuccess;
}
hipError_t launch_cov(const Dims& d, const Args& a, int mode, const double* P0, double* Pout, hipStream_t) {
    MHE_DISPATCH(d.NX, run_waves(d.nwaves, 0, [&](EmuWave& w, int wv, double*) { cov_body<EmuWave, NX>(w, d, a, mode, P0, Pout, wv); }));
    return hipSuccess;
}
hipError_t launch_step(const Dims& d, const Args& a, hipStream_t) {
    MHE_DISPATCH(d.NX, run_waves(d.nwaves, step_lds_doubles(d.NX), [&](EmuWave& w, int wv, double* sm) { step_body<EmuWave, NX, 15u>(w, d, a, wv, sm); }));
    return hipSuccess;
}
}  // namespace mhe

hipError_t launch_step_small(const Dims& d, const Model& m, const StepIO& io, hipStream_t) {
    using namespace mhe;
    const int grid = (d.B + SMALL_GPW - 1) / SMALL_GPW, NXv = 4 * ((d.nZ + 3) / 4);
    if (small_has_y(d)) {
        switch (small_row_slots(d)) {
            case 2: MHE_DISPATCH(NXv, run_waves(grid, small_lds_doubles(d, true), [&](EmuWave& w, int wv, double* sm) { step_small_body<EmuWave, NX, 2>(w, d, m, io, wv, sm); })); break;
            case 3: MHE_DISPATCH(NXv, run_waves(grid, small_lds_doubles(d, true), [&](EmuWave& w, int wv, double* sm) { step_small_body<EmuWave, NX, 3>(w, d, m, io, wv, sm); })); break;
            default: MHE_DISPATCH(NXv, run_waves(grid, small_lds_doubles(d, true), [&](EmuWave& w, int wv, double* sm) { step_small_body<EmuWave, NX, 4>(w, d, m, io, wv, sm); })); break;
        }
    } else {
        // the product runs the variant with the active-set polish on grids beyond one wavefront per SIMD; here: unless
        // MPCQP_EMU_SMALL_POLISH=0 (the tests run both)
        const char* e = getenv("MPCQP_EMU_SMALL_POLISH");
        if (e && e[0] == '0') {
            MHE_DISPATCH(NXv, run_waves(grid, small_lds_doubles(d), [&](EmuWave& w, int wv, double* sm) { step_small_body<EmuWave, NX, 0>(w, d, m, io, wv, sm); }));
        } else {
            MHE_DISPATCH(NXv, run_waves(grid, small_lds_doubles(d), [&](EmuWave& w, int wv, double* sm) { step_small_body<EmuWave, NX, 0, true>(w, d, m, io, wv, sm); }));
        }
    }
    return hipSuccess;
}

namespace mhe {

int waves_for(int, int B, int) {
    const int groups = (B + GPW - 1) / GPW;
    return groups < 2 ? groups : 2;       // two "persistent" wavefronts: the grid-stride loop is exercised
}

}  // namespace mhe
}  // namespace mpcqp
