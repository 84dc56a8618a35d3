// TEST INFRASTRUCTURE ONLY: the 64 lanes of an emulated wavefront as cooperative fibers on ONE host thread.
// A wave-level operation is a barrier; with one OS thread per lane (the first version) every barrier was a futex round
// trip of 64 threads on a handful of cores, and the emulator tests spent most of their time in the kernel.  Here a lane
// that reaches a barrier switches to the next lane (round robin): when the last lane arrives control returns to lane 0,
// which leaves the barrier -- the same semantics as std::barrier for code in which every lane passes the same sequence
// of barriers (which the real barrier needs as well).  The switch is a dozen instructions (callee-saved registers + stack
// pointer, x86-64 System V), no system call.
#pragma once
#include <sys/mman.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>

extern "C" void mpcqp_emu_switch(void** save_sp, void* load_sp);
#ifdef MPCQP_EMU_FIBER_IMPL
__asm__(
    ".text\n"
    ".globl mpcqp_emu_switch\n"
    ".type mpcqp_emu_switch,@function\n"
    "mpcqp_emu_switch:\n"
    "    pushq %rbp\n    pushq %rbx\n    pushq %r12\n    pushq %r13\n    pushq %r14\n    pushq %r15\n"
    "    movq %rsp, (%rdi)\n"
    "    movq %rsi, %rsp\n"
    "    popq %r15\n    popq %r14\n    popq %r13\n    popq %r12\n    popq %rbx\n    popq %rbp\n"
    "    ret\n"
    ".size mpcqp_emu_switch,.-mpcqp_emu_switch\n");
#endif

namespace mpcqp {

class LaneFibers {
public:
    static constexpr int N = 64;
    static constexpr size_t STACK = 1u << 20;          // per lane
    LaneFibers() {
        mem_ = (char*)mmap(nullptr, N * STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (mem_ == (char*)MAP_FAILED) { perror("emu fibers: mmap"); abort(); }
    }
    ~LaneFibers() { munmap(mem_, N * STACK); }
    // run fn(lane) for the N lanes; returns when all of them have returned
    void run(const std::function<void(int)>& fn) {
        fn_ = &fn;
        done_ = 0;
        for (int i = 0; i < N; ++i) {
            uintptr_t top = ((uintptr_t)(mem_ + (size_t)(i + 1) * STACK)) & ~(uintptr_t)15;
            void** p = (void**)(top - 64);             // six register slots, the entry address, one pad: p % 16 == 0
            for (int k = 0; k < 6; ++k) p[k] = nullptr;
            p[6] = (void*)&LaneFibers::entry;
            p[7] = nullptr;
            sp_[i] = p;
        }
        cur_ = 0;
        current() = this;
        mpcqp_emu_switch(&main_sp_, sp_[0]);
    }
    // barrier of the N lanes (called by the running lane)
    void arrive_and_wait() {
        const int me = cur_, nx = (me + 1) % N;
        cur_ = nx;
        mpcqp_emu_switch(&sp_[me], sp_[nx]);
    }

private:
    static LaneFibers*& current() { static thread_local LaneFibers* c = nullptr; return c; }
    static void entry() {
        LaneFibers* f = current();
        const int me = f->cur_;
        (*f->fn_)(me);
        // every lane has passed the same barriers: the lanes after this one are suspended in their last barrier (or
        // not started when there was none) and finish in turn; the last one hands control back to run()
        if (++f->done_ == N) {
            void* dummy;
            mpcqp_emu_switch(&dummy, f->main_sp_);
        }
        const int nx = (me + 1) % N;
        f->cur_ = nx;
        void* dummy;
        mpcqp_emu_switch(&dummy, f->sp_[nx]);
        abort();                                       // (a finished lane is never resumed)
    }
    char* mem_;
    void* sp_[N];
    void* main_sp_ = nullptr;
    const std::function<void(int)>* fn_ = nullptr;
    int cur_ = 0, done_ = 0;
};

// The fibers run in the order 0..63 between barriers; which LANE a fiber plays is a permutation chosen by
// MPCQP_EMU_LANE_ORDER (unset / "forward": identity, "reverse": 63 - i, "random": a fixed pseudo-random permutation), so
// that a missing w.sync() where one lane reads what another wrote in the same section shows up in at least one order
// (with the identity alone a read of a LOWER lane's fresh value always passed).
inline void emu_lane_order(int* perm) {
    for (int i = 0; i < 64; ++i) perm[i] = i;
    if (const char* o = getenv("MPCQP_EMU_LANE_ORDER")) {
        if (o[0] == 'r' && o[1] == 'e') { for (int i = 0; i < 64; ++i) perm[i] = 63 - i; }
        else if (o[0] == 'r' && o[1] == 'a') {
            unsigned x = 2463534242u;
            for (int i = 63; i > 0; --i) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; const int j = (int)(x % (unsigned)(i + 1)); const int t = perm[i]; perm[i] = perm[j]; perm[j] = t; }
        }
    }
}

// the fibers of this host thread (stacks are mapped once)
inline LaneFibers& lane_fibers() { static thread_local LaneFibers f; return f; }

}  // namespace mpcqp
