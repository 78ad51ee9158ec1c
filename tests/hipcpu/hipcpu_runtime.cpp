// TEST INFRASTRUCTURE (see include/hip/hip_runtime.h): the fiber scheduler and the
// synchronous runtime API of the HIP-on-CPU execution model.
#include <hip/hip_runtime.h>

#include <execinfo.h>
#include <signal.h>
#include <sys/mman.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <mutex>
#include <vector>

namespace hipcpu {

LaneView g_lane;

namespace {

enum State { READY, WAIT_BAR, WAIT_WAVE, DONE };

// Context switch without system calls (swapcontext saves the signal mask on every switch): the callee-saved
// registers of the System V x86-64 ABI are pushed on the outgoing stack, the stack pointers are exchanged.
extern "C" void hipcpu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl hipcpu_switch
.type hipcpu_switch,@function
hipcpu_switch:
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
.size hipcpu_switch,.-hipcpu_switch
)");

struct Fiber {
    void* sp;
    State state;
    hipcpu_uint3 tid;
    int lin;
};

struct Wave {
    uint64_t alive, arrived, gen;
    uint64_t buf[2][64];
    uint64_t mask[2];
    int kind[64];
};

constexpr size_t STACK_BYTES = 256 * 1024;
constexpr int MAX_LANES = 1024;

struct Block {
    std::vector<Fiber> f;
    std::vector<Wave> w;
    int n = 0, alive = 0, bar_arrived = 0;
    int cur = -1;
    void* sched_sp = nullptr;
    void (*entry)(void*) = nullptr;
    void* closure = nullptr;
    char* stacks = nullptr;
    hipcpu_uint3 bid;
    dim3 bdim, gdim;
};

Block g_b;
std::recursive_mutex g_launch_lock;

void enter(int i) {
    Fiber& f = g_b.f[i];
    g_b.cur = i;
    g_lane.tid = f.tid;
    g_lane.bid = g_b.bid;
    g_lane.bdim = g_b.bdim;
    g_lane.gdim = g_b.gdim;
}

void yield_to_scheduler() {
    const int me = g_b.cur;
    hipcpu_switch(&g_b.f[me].sp, g_b.sched_sp);
    enter(me);   // resumed
}

void try_release_barrier() {
    if (g_b.alive > 0 && g_b.bar_arrived == g_b.alive) {
        for (int i = 0; i < g_b.n; ++i)
            if (g_b.f[i].state == WAIT_BAR) g_b.f[i].state = READY;
        g_b.bar_arrived = 0;
    }
}

// `partial`: every live work-item of the workgroup is blocked and the lanes of this wavefront that have not
// arrived sit at a barrier (or have diverged for good): the collective executes with the lanes that did
// arrive, as it does on the hardware under a partial exec mask.
void try_release_wave(int wi, bool partial = false) {
    Wave& w = g_b.w[wi];
    if (w.arrived != 0 && (partial || w.arrived == w.alive)) {
        for (int l = 0; l < 64; ++l)
            if (((w.arrived >> l) & 1ull) && w.kind[l] != w.kind[__builtin_ctzll(w.arrived)]) {
                std::fprintf(stderr, "[hipcpu] lanes of one wavefront met in different kinds of wave collectives "
                             "(divergent branches): not supported by this execution model\n");
                std::abort();
            }
        w.mask[w.gen & 1] = w.arrived;
        w.arrived = 0;
        w.gen++;
        const int lo = wi * 64, hi = std::min(g_b.n, lo + 64);
        for (int i = lo; i < hi; ++i)
            if (g_b.f[i].state == WAIT_WAVE) g_b.f[i].state = READY;
    }
}

void trampoline() {
    g_b.entry(g_b.closure);
    const int me = g_b.cur;
    Fiber& f = g_b.f[me];
    f.state = DONE;
    g_b.alive--;
    const int wi = me / 64;
    g_b.w[wi].alive &= ~(1ull << (me & 63));
    try_release_barrier();
    try_release_wave(wi);
    hipcpu_switch(&f.sp, g_b.sched_sp);   // never resumed
    std::abort();
}

}  // namespace

void sync_threads() {
    const int me = g_b.cur;
    g_b.f[me].state = WAIT_BAR;
    g_b.bar_arrived++;
    try_release_barrier();
    if (g_b.f[me].state != READY) yield_to_scheduler();
}

const uint64_t* wave_exchange(int kind, uint64_t v, uint64_t* mask, int* lane) {
    const int me = g_b.cur;
    const int wi = me / 64, l = me & 63;
    Wave& w = g_b.w[wi];
    const int par = (int)(w.gen & 1);
    w.buf[par][l] = v;
    w.kind[l] = kind;
    w.arrived |= 1ull << l;
    g_b.f[me].state = WAIT_WAVE;
    try_release_wave(wi);
    if (g_b.f[me].state != READY) yield_to_scheduler();
    *mask = w.mask[par];
    *lane = l;
    return w.buf[par];
}

// HIPCPU_BACKTRACE=1: a fault inside a kernel (a guard page, say) prints the faulting address, the work-item and the
// frames of its fiber before the process dies (symbolise with addr2line -e libwarpx_amd_hipcpu.so <offsets>)
static void fault_handler(int sig, siginfo_t* info, void*) {
    char buf[256];
    int n = std::snprintf(buf, sizeof(buf), "[hipcpu] signal %d at address %p in workgroup (%u,%u,%u) work-item %d\n", sig, info->si_addr,
                          g_b.bid.x, g_b.bid.y, g_b.bid.z, g_b.cur);
    if (write(2, buf, (size_t)n) < 0) {}
    void* frames[48];
    const int nf = backtrace(frames, 48);
    backtrace_symbols_fd(frames, nf, 2);
    _exit(139);
}

void run_grid(dim3 grid, dim3 block, void (*entry)(void*), void* closure, const char* name) {
    std::lock_guard<std::recursive_mutex> lock(g_launch_lock);
    static const bool bt = [] {
        if (!std::getenv("HIPCPU_BACKTRACE")) return false;
        static char altstack[1 << 16];
        stack_t ss{};
        ss.ss_sp = altstack; ss.ss_size = sizeof(altstack);
        sigaltstack(&ss, nullptr);
        struct sigaction sa{};
        sa.sa_sigaction = fault_handler;
        sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
        sigaction(SIGSEGV, &sa, nullptr);
        sigaction(SIGBUS, &sa, nullptr);
        return true;
    }();
    (void)bt;
    static const bool trace = std::getenv("HIPCPU_TRACE") != nullptr;   // the last line names a crashing kernel
    if (trace)
        std::fprintf(stderr, "[hipcpu] %s grid (%u,%u,%u) block (%u,%u,%u)\n", name, grid.x, grid.y, grid.z, block.x,
                     block.y, block.z);
    const long n = (long)block.x * block.y * block.z;
    if (n <= 0 || n > MAX_LANES) {
        std::fprintf(stderr, "[hipcpu] %s: workgroup of %ld work-items (limit %d)\n", name, n, MAX_LANES);
        std::abort();
    }
    if ((long)grid.x * grid.y * grid.z <= 0) {
        std::fprintf(stderr, "[hipcpu] %s: empty grid (hipErrorInvalidConfiguration on the device)\n", name);
        std::abort();
    }
    Block& b = g_b;
    if (!b.stacks) {
        b.stacks = static_cast<char*>(mmap(nullptr, STACK_BYTES * MAX_LANES, PROT_READ | PROT_WRITE,
                                           MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0));
        if (b.stacks == MAP_FAILED) { std::perror("[hipcpu] mmap"); std::abort(); }
    }
    b.n = (int)n;
    b.f.resize(n);
    b.w.resize((n + 63) / 64);
    b.entry = entry;
    b.closure = closure;
    b.bdim = block;
    b.gdim = grid;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                b.bid = {bx, by, bz};
                b.alive = (int)n;
                b.bar_arrived = 0;
                for (auto& w : b.w) { w.alive = 0; w.arrived = 0; w.gen = 0; }
                int lin = 0;
                for (unsigned tz = 0; tz < block.z; ++tz)
                    for (unsigned ty = 0; ty < block.y; ++ty)
                        for (unsigned tx = 0; tx < block.x; ++tx, ++lin) {
                            Fiber& f = b.f[lin];
                            f.state = READY;
                            f.tid = {tx, ty, tz};
                            f.lin = lin;
                            // a fresh stack that "returns" into trampoline with rsp = 8 mod 16
                            void** top = reinterpret_cast<void**>(b.stacks + STACK_BYTES * (lin + 1));
                            top[-1] = nullptr;
                            top[-2] = reinterpret_cast<void*>(&trampoline);
                            for (int r = 3; r <= 8; ++r) top[-r] = nullptr;
                            f.sp = top - 8;
                            b.w[lin / 64].alive |= 1ull << (lin & 63);
                        }
                while (b.alive > 0) {
                    bool progressed = false;
                    for (int i = 0; i < b.n; ++i) {
                        if (b.f[i].state != READY) continue;
                        enter(i);
                        hipcpu_switch(&b.sched_sp, b.f[i].sp);
                        progressed = true;
                    }
                    if (!progressed) {
                        for (size_t wi = 0; wi < b.w.size(); ++wi)
                            if (b.w[wi].arrived != 0) { try_release_wave((int)wi, true); progressed = true; }
                    }
                    if (!progressed) {
                        int nb = 0, nw = 0;
                        for (int i = 0; i < b.n; ++i) { nb += b.f[i].state == WAIT_BAR; nw += b.f[i].state == WAIT_WAVE; }
                        std::fprintf(stderr, "[hipcpu] %s: deadlock in workgroup (%u,%u,%u): %d work-items alive, %d at a "
                                     "barrier, %d at a wave collective (divergent barrier or collective)\n",
                                     name, bx, by, bz, b.alive, nb, nw);
                        std::abort();
                    }
                }
            }
    b.cur = -1;
}

}  // namespace hipcpu

// ---- runtime API ------------------------------------------------------------------------
struct hipcpu_stream { int dummy; };
struct hipcpu_event { std::chrono::steady_clock::time_point t; };

// HIPCPU_GUARD_PAGES=1: every allocation ends on an inaccessible page and starts behind one, so that a kernel
// reading or writing outside a buffer it was handed faults instead of touching a neighbour allocation (on the
// device such an access is silent as long as it stays inside mapped memory).
namespace {
const bool g_guard_pages = std::getenv("HIPCPU_GUARD_PAGES") != nullptr;
constexpr size_t PAGE = 4096;
struct GuardHeader { void* base; size_t total; };
}
hipError_t hipMalloc(void** p, size_t bytes) {
    if (bytes == 0) bytes = 8;
    if (g_guard_pages) {
        const size_t body = (bytes + 255) / 256 * 256;                      // keeps hipMalloc's 256-B alignment
        const size_t inner = (body + sizeof(GuardHeader) + PAGE - 1) / PAGE * PAGE;
        const size_t total = inner + 2 * PAGE;
        char* base = static_cast<char*>(mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
        if (base == MAP_FAILED) return hipErrorOutOfMemory;
        char* user = base + PAGE + inner - body;                            // the buffer ends where the last page starts
        GuardHeader* h = reinterpret_cast<GuardHeader*>(base + PAGE);
        if (reinterpret_cast<char*>(h + 1) > user) { munmap(base, total); return hipErrorOutOfMemory; }
        *h = {base, total};
        mprotect(base, PAGE, PROT_NONE);
        mprotect(base + PAGE + inner, PAGE, PROT_NONE);
        *p = user;
        return hipSuccess;
    }
    void* q = nullptr;
    if (posix_memalign(&q, 256, bytes) != 0) return hipErrorOutOfMemory;
    *p = q;
    return hipSuccess;
}
hipError_t hipFree(void* p) {
    if (!p) return hipSuccess;
    if (g_guard_pages) {
        char* page = reinterpret_cast<char*>(reinterpret_cast<uintptr_t>(p) / PAGE * PAGE);
        // the header sits at the start of the first accessible page of the mapping: walk back to it
        while (true) {
            GuardHeader* h = reinterpret_cast<GuardHeader*>(page);
            if (h->base == page - PAGE && h->total >= 3 * PAGE && (h->total % PAGE) == 0) { munmap(h->base, h->total); break; }
            page -= PAGE;
        }
        return hipSuccess;
    }
    std::free(p);
    return hipSuccess;
}
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) std::memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t) { return hipMemcpy(d, s, n, k); }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { if (n) std::memset(d, v, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { if (n) std::memset(d, v, n); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipcpu error"; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new hipcpu_stream{0}; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipcpu_event{std::chrono::steady_clock::now()}; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
