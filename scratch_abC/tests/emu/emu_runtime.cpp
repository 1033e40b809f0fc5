// TEST INFRASTRUCTURE ONLY: fiber scheduler behind tests/emu/include/hip/hip_runtime.h.
// Each HIP thread of a workgroup is a fiber with its own stack; fibers yield only at
// __syncthreads() and at wave collectives (shuffle / MFMA).  A workgroup runs start to
// finish on ONE host thread; the workgroups of a launch are handed out to a small pool
// of host threads (LBC_EMU_THREADS, default min(8, cores); 1 = the calling thread only,
// workgroups in index order).  Everything a workgroup touches in here, the built-in
// index variables and every __shared__ array (static thread_local under the emulator)
// is per host thread.  The kernels use no atomics and no inter-workgroup ordering, so
// results do not depend on the thread count.
// LBC_EMU_ORDER=1 reverses the lane visiting order inside a workgroup, which
// changes the interleaving between rendezvous points and exposes missing
// barriers (run the suite under both orders).
#include <hip/hip_runtime.h>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

thread_local emu_uint3 threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;

extern "C" void emu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
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
)");

namespace emu {
namespace {
enum State { READY = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };
struct Fiber {
    void* sp = nullptr;
    State state = READY;
    emu_uint3 tid{0, 0, 0};
    int linear = 0;
    int coll = 0;   // number of wave collectives this lane has entered
};
constexpr size_t kStack = 256 * 1024;
thread_local char* g_stacks = nullptr;
thread_local size_t g_nstacks = 0;
thread_local std::vector<Fiber> g_fibers;
thread_local Fiber* g_cur = nullptr;
thread_local void* g_sched_sp = nullptr;
thread_local const std::function<void()>* g_body = nullptr;
thread_local std::vector<Slot> g_slots;   // [wave][parity][64]

void yield_to_scheduler() { emu_switch(&g_cur->sp, g_sched_sp); }

void fiber_main() {
    (*g_body)();
    g_cur->state = DONE;
    yield_to_scheduler();
    fprintf(stderr, "emu: resumed a finished fiber\n");
    abort();
}
}  // namespace

int lane_id() { return g_cur->linear & 63; }

void sync_block() {
    g_cur->state = WAIT_BLOCK;
    yield_to_scheduler();
}

Slot* wave_slots_begin() {
    int wave = g_cur->linear >> 6;
    int parity = g_cur->coll & 1;
    return &g_slots[(size_t)(wave * 2 + parity) * 64];
}

void wave_rendezvous() {
    g_cur->coll++;
    g_cur->state = WAIT_WAVE;
    yield_to_scheduler();
}

namespace {
// one workgroup, start to finish, on the calling host thread
void run_block(dim3 grid, dim3 block, unsigned bx, unsigned by, unsigned bz, const std::function<void()>& body, bool reverse) {
    const int nthreads = (int)(block.x * block.y * block.z);
    if ((size_t)nthreads > g_nstacks) {
        free(g_stacks);
        g_stacks = (char*)aligned_alloc(64, kStack * (size_t)nthreads);
        g_nstacks = (size_t)nthreads;
    }
    const int nwaves = (nthreads + 63) / 64;
    g_slots.assign((size_t)nwaves * 2 * 64, Slot{});
    blockDim = block;
    gridDim = grid;
    g_body = &body;
    blockIdx = {bx, by, bz};
    g_fibers.assign((size_t)nthreads, Fiber{});
    for (int t = 0; t < nthreads; ++t) {
        Fiber& f = g_fibers[(size_t)t];
        f.linear = t;
        f.tid.x = (unsigned)t % block.x;
        f.tid.y = ((unsigned)t / block.x) % block.y;
        f.tid.z = (unsigned)t / (block.x * block.y);
        char* top = g_stacks + kStack * (size_t)(t + 1);
        void** sp = (void**)top;
        *--sp = nullptr;                 // fake return address for fiber_main
        *--sp = (void*)&fiber_main;      // 'ret' target of the first switch
        for (int i = 0; i < 6; ++i) *--sp = nullptr;   // rbp rbx r12..r15
        f.sp = sp;
    }
    int ndone = 0;
    while (ndone < nthreads) {
        bool progressed = false;
        for (int i = 0; i < nthreads; ++i) {
            int t = reverse ? nthreads - 1 - i : i;
            Fiber& f = g_fibers[(size_t)t];
            if (f.state != READY) continue;
            g_cur = &f;
            threadIdx = f.tid;
            emu_switch(&g_sched_sp, f.sp);
            progressed = true;
            if (f.state == DONE) ++ndone;
        }
        // block barrier: every live fiber waits on it
        int nblk = 0, nlive = 0;
        for (auto& f : g_fibers) { if (f.state != DONE) ++nlive; if (f.state == WAIT_BLOCK) ++nblk; }
        if (nlive > 0 && nblk == nlive) {
            for (auto& f : g_fibers) if (f.state == WAIT_BLOCK) f.state = READY;
            progressed = true;
        }
        // wave rendezvous: all 64 lanes (or all lanes of a partial last wave) must arrive
        for (int w = 0; w < nwaves; ++w) {
            int lo = w * 64, hi = lo + 64 < nthreads ? lo + 64 : nthreads;
            int nw = 0, ndead = 0;
            for (int t = lo; t < hi; ++t) {
                if (g_fibers[(size_t)t].state == WAIT_WAVE) ++nw;
                if (g_fibers[(size_t)t].state == DONE) ++ndead;
            }
            if (nw > 0 && nw + ndead == hi - lo) {
                if (ndead) { fprintf(stderr, "emu: wave collective with exited lanes (block %u,%u,%u wave %d)\n", bx, by, bz, w); abort(); }
                for (int t = lo; t < hi; ++t) g_fibers[(size_t)t].state = READY;
                progressed = true;
            }
        }
        if (!progressed) {
            fprintf(stderr, "emu: deadlock in block (%u,%u,%u): divergent barrier/collective\n", bx, by, bz);
            abort();
        }
    }
    g_body = nullptr;
}

// The pool: workers sleep between launches; a launch publishes one job, everybody (the caller included) pulls workgroup
// indices from an atomic counter until the grid is exhausted.  Never destroyed (workers may be parked at process exit).
struct Pool {
    std::mutex m;
    std::condition_variable wake, done;
    std::vector<std::thread> workers;
    unsigned long long job = 0;          // generation
    int busy = 0;
    dim3 grid, block;
    const std::function<void()>* body = nullptr;
    bool reverse = false;
    std::atomic<unsigned long long> next{0};
    unsigned long long total = 0;

    void drain() {
        for (;;) {
            const unsigned long long i = next.fetch_add(1, std::memory_order_relaxed);
            if (i >= total) return;
            const unsigned bx = (unsigned)(i % grid.x), by = (unsigned)((i / grid.x) % grid.y), bz = (unsigned)(i / ((unsigned long long)grid.x * grid.y));
            run_block(grid, block, bx, by, bz, *body, reverse);
        }
    }
    void worker() {
        unsigned long long seen = 0;
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            wake.wait(lk, [&] { return job != seen; });
            seen = job;
            lk.unlock();
            drain();
            lk.lock();
            if (--busy == 0) done.notify_all();
        }
    }
    void run(dim3 g, dim3 b, const std::function<void()>& f, bool rev, int nthreads) {
        total = (unsigned long long)g.x * g.y * g.z;
        grid = g; block = b; body = &f; reverse = rev;
        next.store(0, std::memory_order_relaxed);
        int helpers = nthreads - 1;
        if ((unsigned long long)helpers > total - 1) helpers = (int)(total - 1);
        if (helpers > 0) {
            std::unique_lock<std::mutex> lk(m);
            while ((int)workers.size() < nthreads - 1) workers.emplace_back([this] { worker(); });
            // every parked worker wakes for a generation (those beyond `helpers` find the counter exhausted quickly)
            busy = (int)workers.size();
            ++job;
            wake.notify_all();
        }
        drain();
        if (helpers > 0) {
            std::unique_lock<std::mutex> lk(m);
            done.wait(lk, [&] { return busy == 0; });
        }
        body = nullptr;
    }
};

int pool_threads() {
    static const int n = [] {
        const char* e = getenv("LBC_EMU_THREADS");
        int v = e ? atoi(e) : 0;
        if (v <= 0) {
            v = (int)std::thread::hardware_concurrency();
            if (v > 8) v = 8;
        }
        return v < 1 ? 1 : v;
    }();
    return n;
}
}  // namespace

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    const int nthreads = (int)(block.x * block.y * block.z);
    if (nthreads <= 0 || nthreads > 1024) { fprintf(stderr, "emu: bad block size %d\n", nthreads); abort(); }
    if (grid.x == 0 || grid.y == 0 || grid.z == 0) return;
    const char* ord = getenv("LBC_EMU_ORDER");
    const bool reverse = ord && ord[0] == '1';
    static Pool* pool = new Pool;     // one launch at a time per process: the kernels are enqueued from one host thread
    static std::mutex launch_mutex;
    std::lock_guard<std::mutex> g(launch_mutex);
    pool->run(grid, block, body, reverse, pool_threads());
}
}  // namespace emu
