// TEST-ONLY: storage and the fiber scheduler of the emulator (see hip_emu.h).
#define LV_EMU_IMPL
#include "hip_emu.h"

// void lv_emu_switch(void** save_sp, void* load_sp): System V x86-64, callee-saved registers only
asm(R"(
.text
.globl lv_emu_switch
.type lv_emu_switch,@function
lv_emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size lv_emu_switch,.-lv_emu_switch
)");

namespace lv_emu {

State g_state;
lv_emu_idx t_threadIdx{0, 0, 0};
lv_emu_idx t_blockIdx{0, 0, 0};
int t_lin = 0;

// first frame of every fiber: run the kernel body for this thread, then hand the OS thread on for good
static void fiber_main() {
    State& s = st();
    enter(s.fibers[s.cur]);
    s.job();
    Fiber& f = s.fibers[s.cur];
    f.done = true;
    --s.remaining;
    bar_leave(f.blk->bar);
    bar_leave(f.blk->waves[f.lin / kWave].bar);
    // unlink from the ring of unfinished fibers; the successor runs next
    const int me = s.cur, nxt = s.ring_next[me], prv = s.ring_prev[me];
    s.ring_next[prv] = nxt;
    s.ring_prev[nxt] = prv;
    const int nx = nxt != me ? nxt : -1;
    void* dummy;
    if (nx >= 0) { s.cur = nx; lv_emu_switch(&dummy, s.fibers[nx].sp); }
    else { s.cur = -1; lv_emu_switch(&dummy, s.main_sp); }
    __builtin_trap();                      // a finished fiber is never resumed
}

void run_live(int nlive) {
    State& s = st();
    s.nlive = nlive;
    s.remaining = nlive;
    for (int i = 0; i < nlive; ++i) {
        Fiber& f = s.fibers[i];
        f.done = false;
        // initial frame: six callee-saved slots, then fiber_main as the return address at a 16-byte aligned slot
        // (so that rsp % 16 == 8 on entry, as after a call), then a null return address above it
        uintptr_t top = ((uintptr_t)f.stack + kStack - 64) & ~(uintptr_t)15;
        void** sp = (void**)top;
        sp[1] = nullptr;
        sp[0] = (void*)&fiber_main;
        for (int k = 1; k <= 6; ++k) sp[-k] = nullptr;
        f.sp = (void*)(sp - 6);
    }
    s.ring_next.resize((size_t)nlive);
    s.ring_prev.resize((size_t)nlive);
    for (int i = 0; i < nlive; ++i) { s.ring_next[i] = (i + 1) % nlive; s.ring_prev[i] = (i + nlive - 1) % nlive; }
    s.cur = 0;
    lv_emu_switch(&s.main_sp, s.fibers[0].sp);
    if (s.remaining != 0) { fprintf(stderr, "lv_emu: %d fibers never finished\n", s.remaining); abort(); }
}

}  // namespace lv_emu
