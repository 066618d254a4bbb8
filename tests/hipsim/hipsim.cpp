// hipsim runtime: fibers-per-HIP-thread execution of kernels on the host.  TEST INFRASTRUCTURE ONLY
// (see tests/hipsim/hip/hip_runtime.h).  Single OS thread; blocks run sequentially; the threads of a block are
// ucontext fibers scheduled round-robin and switch only at __syncthreads() / wave shuffles.
#include <hip/hip_runtime.h>

#include <ucontext.h>

#include <chrono>
#include <vector>

namespace hipsim {

ThreadState* cur = nullptr;
alignas(64) unsigned char dyn_smem[160 * 1024];
dim3 cur_block, cur_bdim, cur_gdim;

namespace {
constexpr size_t kStack = 256 * 1024;

struct Fiber {
    ThreadState st;
    ucontext_t ctx;
    bool done = false;
    int linear = 0;
};

struct Wave {
    uint32_t slot[64];
    uint32_t slot2[64];
    int alive = 0, arrived = 0;
    unsigned gen = 0;
};

struct BlockRun {
    std::vector<Fiber> fibers;
    std::vector<Wave> waves;
    int alive = 0, arrived = 0;
    unsigned gen = 0;
    const std::function<void()>* body = nullptr;
    ucontext_t sched;
    Fiber* running = nullptr;
};

BlockRun* g_run = nullptr;
unsigned long g_events = 0;   // barrier arrivals/releases + fiber completions (deadlock detector)
std::vector<char*> g_stacks;

void yield_to_sched() {
    Fiber* f = g_run->running;
    swapcontext(&f->ctx, &g_run->sched);
}

void release_block_if_complete(BlockRun* r) {
    if (r->alive > 0 && r->arrived == r->alive) {
        r->arrived = 0;
        r->gen++;
    }
}
void release_wave_if_complete(Wave* w) {
    if (w->alive > 0 && w->arrived == w->alive) {
        w->arrived = 0;
        w->gen++;
    }
}

void trampoline() {
    BlockRun* r = g_run;
    Fiber* f = r->running;
    (*r->body)();
    f->done = true;
    ++g_events;
    r->alive--;
    Wave* w = &r->waves[f->linear / 64];
    w->alive--;
    release_block_if_complete(r);   // exited threads no longer take part in barriers
    release_wave_if_complete(w);
    swapcontext(&f->ctx, &r->sched);
}

void wave_barrier(Wave* w) {
    const unsigned gen = w->gen;
    ++g_events;
    w->arrived++;
    release_wave_if_complete(w);
    while (w->gen == gen) yield_to_sched();
}
}  // namespace

void block_barrier() {
    BlockRun* r = g_run;
    const unsigned gen = r->gen;
    ++g_events;
    r->arrived++;
    release_block_if_complete(r);
    while (r->gen == gen) yield_to_sched();
}

int lane_id() { return g_run->running->linear & 63; }

uint32_t wave_exchange(uint32_t v, int src_lane) {
    BlockRun* r = g_run;
    Fiber* f = r->running;
    Wave* w = &r->waves[f->linear / 64];
    w->slot[f->linear & 63] = v;
    wave_barrier(w);
    const uint32_t out = w->slot[src_lane & 63];
    wave_barrier(w);
    return out;
}

// every lane deposits two words; after one rendezvous every lane sees all 64 pairs (used by the MFMA emulation, which
// would otherwise need 32 separate exchanges per instruction)
void wave_allgather2(uint32_t a, uint32_t b, uint32_t* out_a, uint32_t* out_b) {
    BlockRun* r = g_run;
    Fiber* f = r->running;
    Wave* w = &r->waves[f->linear / 64];
    w->slot[f->linear & 63] = a;
    w->slot2[f->linear & 63] = b;
    wave_barrier(w);
    for (int i = 0; i < 64; ++i) { out_a[i] = w->slot[i]; out_b[i] = w->slot2[i]; }
    wave_barrier(w);
}

void run_grid(dim3 grid, dim3 block, const std::function<void()>& body) {
    const int nthreads = (int)(block.x * block.y * block.z);
    if (nthreads <= 0 || nthreads > 1024) {
        fprintf(stderr, "hipsim: bad block size %d\n", nthreads);
        abort();
    }
    while ((int)g_stacks.size() < nthreads) g_stacks.push_back((char*)malloc(kStack));
    BlockRun run;
    run.body = &body;
    g_run = &run;
    cur_bdim = block;
    cur_gdim = grid;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                cur_block = dim3(bx, by, bz);
                run.fibers.assign(nthreads, Fiber());
                run.waves.assign((nthreads + 63) / 64, Wave());
                run.alive = nthreads;
                run.arrived = 0;
                run.gen = 0;
                for (int i = 0; i < nthreads; ++i) {
                    Fiber& f = run.fibers[i];
                    f.linear = i;
                    f.st.tid = dim3(i % block.x, (i / block.x) % block.y, i / (block.x * block.y));
                    run.waves[i / 64].alive++;
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = g_stacks[i];
                    f.ctx.uc_stack.ss_size = kStack;
                    f.ctx.uc_link = nullptr;
                    makecontext(&f.ctx, trampoline, 0);
                }
                int remaining = nthreads;
                while (remaining > 0) {
                    const unsigned long ev0 = g_events;
                    for (int i = 0; i < nthreads; ++i) {
                        Fiber& f = run.fibers[i];
                        if (f.done) continue;
                        run.running = &f;
                        cur = &f.st;
                        swapcontext(&run.sched, &f.ctx);
                        if (f.done) --remaining;
                    }
                    if (g_events == ev0) {
                        fprintf(stderr, "hipsim: deadlock (divergent barrier or shuffle) in block (%u,%u,%u)\n", bx, by, bz);
                        abort();
                    }
                }
            }
    g_run = nullptr;
    cur = nullptr;
}

}  // namespace hipsim

// ---- host runtime API ---------------------------------------------------------------------------
struct hipsimStream { int id; };
struct hipsimEvent { std::chrono::steady_clock::time_point t; };

extern "C" {
hipError_t hipMalloc(void** p, size_t n) {
    *p = nullptr;
    if (posix_memalign(p, 256, n ? n : 1)) return hipErrorOutOfMemory;
    memset(*p, 0xCD, n);   // poison like uninitialised device memory
    return hipSuccess;
}
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return posix_memalign(p, 256, n ? n : 1) ? hipErrorOutOfMemory : hipSuccess; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t* s) { *s = new hipsimStream{1}; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new hipsimStream{1}; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipPeekAtLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipsim error"; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    memset(p, 0, sizeof *p);
    snprintf(p->name, sizeof p->name, "hipsim host simulator");
    snprintf(p->gcnArchName, sizeof p->gcnArchName, "hipsim");
    p->multiProcessorCount = 1;
    p->clockRate = 1000000;
    return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipsimEvent(); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new hipsimEvent(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
// Stream capture / graphs are not simulated: the engine must fall back to plain launches.
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorNotSupported; }
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return hipErrorNotSupported; }
hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t, void*, void*, size_t) { *e = nullptr; return hipErrorNotSupported; }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }
hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
}
