// hipemu runtime: runs the blocks of a launch one after the other; the threads of a block are fibers (hand-rolled
// x86-64 context switch) scheduled round-robin between synchronisation points.  TEST INFRASTRUCTURE ONLY.
#include <stdio.h>
#include <sys/mman.h>
#include <vector>
#include "hip/hip_runtime.h"

// AddressSanitizer build (tools/hipemu/build.py --asan): the fiber switches are announced to the runtime, which otherwise takes a
// thread that continues on another stack for a stack overflow, and a re-used fiber stack is unpoisoned before its next thread.
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define HIPEMU_ASAN 1
#include <sanitizer/asan_interface.h>
#include <sanitizer/common_interface_defs.h>
#endif
#endif

extern "C" void hipemu_switch(void** save_sp, void* new_sp);
extern "C" void hipemu_fiber_entry();

namespace hipemu {
void fiber_main_export();

dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
int g_lane = 0;

namespace {
enum State { RUNNABLE = 0, AT_WAVE = 1, AT_BLOCK = 2, DONE = 3 };
constexpr size_t STACK = 512 * 1024;
struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    int state = DONE;
    void* fake = nullptr;        // ASAN: the fiber's fake-stack handle while it is switched out
};
#ifdef HIPEMU_ASAN
void* g_sched_fake = nullptr;
const void* g_sched_bottom = nullptr;
size_t g_sched_size = 0;
#endif
std::vector<Fiber> g_fibers;
void* g_sched_sp = nullptr;
int g_cur = -1;
body_fn g_fn = nullptr;
void* g_ctx = nullptr;
std::vector<char> g_smem;
const void* g_wave_slots[64][64];        // [wave][lane] -- up to 4096 threads per block

asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
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
.size hipemu_switch,.-hipemu_switch
)");

void yield_to_scheduler(int state) {
    Fiber& f = g_fibers[g_cur];
    f.state = state;
#ifdef HIPEMU_ASAN
    __sanitizer_start_switch_fiber(state == DONE ? nullptr : &f.fake, g_sched_bottom, g_sched_size);
#endif
    hipemu_switch(&f.sp, g_sched_sp);
#ifdef HIPEMU_ASAN
    __sanitizer_finish_switch_fiber(f.fake, nullptr, nullptr);
#endif
}

struct PendingCopy { const char* src; char* dst; int size; };
std::vector<std::vector<PendingCopy>> g_pending;       // LDS-DMA copies in flight, per thread of the block
bool dma_outstanding(int t) { return t < (int)g_pending.size() && !g_pending[t].empty(); }
void fiber_main() {
#ifdef HIPEMU_ASAN
    __sanitizer_finish_switch_fiber(nullptr, &g_sched_bottom, &g_sched_size);      // first entry: learn the scheduler's stack
#endif
    g_fn(g_ctx);
    if (dma_outstanding(g_cur)) {
        fprintf(stderr, "hipemu: a thread finished with LDS-DMA copies it never waited for\n");
        abort();
    }
    g_wave_slots[g_cur >> 6][g_cur & 63] = nullptr;
    yield_to_scheduler(DONE);
    fprintf(stderr, "hipemu: resumed a finished fiber\n");
    abort();
}

void prepare(Fiber& f) {
    if (!f.stack) {
        f.stack = (char*)mmap(nullptr, STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (f.stack == MAP_FAILED) { perror("hipemu: mmap"); abort(); }
    }
#ifdef HIPEMU_ASAN
    __asan_unpoison_memory_region(f.stack, STACK);   // the previous thread on this stack never unwound its frames
    f.fake = nullptr;
#endif
    uintptr_t top = ((uintptr_t)f.stack + STACK) & ~(uintptr_t)15;
    void** p = (void**)(top - 16);                 // return-address slot (16-byte aligned -> rsp % 16 == 8 at entry)
    p[0] = (void*)&hipemu_fiber_entry;
    p[1] = nullptr;
    void** sp = p - 6;                             // r15 r14 r13 r12 rbx rbp
    for (int i = 0; i < 6; ++i) sp[i] = nullptr;
    f.sp = sp;
    f.state = RUNNABLE;
}

void set_ids(int t) {
    const unsigned bx = g_blockDim.x, by = g_blockDim.y;
    g_threadIdx.x = t % bx;
    g_threadIdx.y = (t / bx) % by;
    g_threadIdx.z = t / (bx * by);
    g_lane = t & 63;
    g_cur = t;
}

void run_block(int nthr) {
    const int nwaves = (nthr + 63) / 64;
    for (int t = 0; t < nthr; ++t) prepare(g_fibers[t]);
    for (int w = 0; w < nwaves; ++w)
        for (int l = 0; l < 64; ++l) g_wave_slots[w][l] = nullptr;
    for (;;) {
        bool ran = false;
        for (int t = 0; t < nthr; ++t) {
            Fiber& f = g_fibers[t];
            if (f.state != RUNNABLE) continue;
            set_ids(t);
#ifdef HIPEMU_ASAN
            __sanitizer_start_switch_fiber(&g_sched_fake, f.stack, STACK);
#endif
            hipemu_switch(&g_sched_sp, f.sp);
#ifdef HIPEMU_ASAN
            __sanitizer_finish_switch_fiber(g_sched_fake, nullptr, nullptr);
#endif
            ran = true;
        }
        // every fiber is now DONE, AT_WAVE or AT_BLOCK
        bool released = false, all_done = true, all_block = true;
        for (int w = 0; w < nwaves; ++w) {
            int n_wave = 0, n_block = 0;
            const int t0 = w * 64, t1 = t0 + 64 < nthr ? t0 + 64 : nthr;
            for (int t = t0; t < t1; ++t) {
                n_wave += g_fibers[t].state == AT_WAVE;
                n_block += g_fibers[t].state == AT_BLOCK;
            }
            if (n_wave && n_block) {
                fprintf(stderr, "hipemu: wave %d diverged: %d lanes at a cross-lane op, %d at __syncthreads\n", w, n_wave, n_block);
                abort();
            }
            if (n_wave) {
                for (int t = t0; t < t1; ++t)
                    if (g_fibers[t].state == AT_WAVE) g_fibers[t].state = RUNNABLE;
                released = true;
                all_done = all_block = false;
            } else if (n_block) {
                all_done = false;
            }
        }
        if (released) continue;
        if (all_done) break;
        if (all_block) {                                        // every live thread reached the barrier
            for (int t = 0; t < nthr; ++t)
                if (g_fibers[t].state == AT_BLOCK) g_fibers[t].state = RUNNABLE;
            continue;
        }
        if (!ran) { fprintf(stderr, "hipemu: deadlock\n"); abort(); }
    }
}
}  // namespace
void fiber_main_export() { fiber_main(); }

void* dyn_smem() { return g_smem.data(); }

void block_sync() { yield_to_scheduler(AT_BLOCK); }
const void* const* wave_publish(const void* mine) {
    g_wave_slots[g_cur >> 6][g_cur & 63] = mine;
    yield_to_scheduler(AT_WAVE);
    return g_wave_slots[g_cur >> 6];
}
void wave_release() { yield_to_scheduler(AT_WAVE); }

// ---- LDS-DMA: deferred copies, per issuing thread (every lane of a wave issues the same sequence) -------------------------------
void dma_issue(const char* gsrc_lane, char* lds_wave_base, int size) {
    const void* const* all = wave_publish(lds_wave_base);
    for (int i = 0; i < 64; ++i)
        if (all[i] && all[i] != (const void*)lds_wave_base) {
            fprintf(stderr, "hipemu: LDS-DMA destination base is not wave-uniform\n");
            abort();
        }
    wave_release();
    char* dst = lds_wave_base + (size_t)g_lane * size;
    if (dst < g_smem.data() || dst + size > g_smem.data() + g_smem.size()) {
        fprintf(stderr, "hipemu: LDS-DMA destination outside the dynamic LDS allocation\n");
        abort();
    }
    memset(dst, 0xFF, size);                            // in flight: whoever reads it now gets NaNs
    if ((int)g_pending.size() <= g_cur) g_pending.resize(g_cur + 1);
    g_pending[g_cur].push_back(PendingCopy{gsrc_lane, dst, size});
}
void dma_wait(int keep) {
    if ((int)g_pending.size() <= g_cur) return;
    auto& q = g_pending[g_cur];
    const int done = (int)q.size() - keep;
    if (done <= 0) return;
    for (int i = 0; i < done; ++i) memcpy(q[i].dst, q[i].src, q[i].size);
    q.erase(q.begin(), q.begin() + done);
}

void launch(dim3 grid, dim3 block, size_t shmem, body_fn fn, void* ctx) {
    const int nthr = (int)(block.x * block.y * block.z);
    if (nthr <= 0 || nthr > 4096) { fprintf(stderr, "hipemu: bad block size %d\n", nthr); abort(); }
    if ((int)g_fibers.size() < nthr) g_fibers.resize(nthr);
    g_smem.assign(shmem + 64, (char)0xFF);                      // NaN-poisoned: reads of unwritten LDS show up
    g_fn = fn;
    g_ctx = ctx;
    g_blockDim = block;
    g_gridDim = grid;
    for (unsigned z = 0; z < grid.z; ++z)
        for (unsigned y = 0; y < grid.y; ++y)
            for (unsigned x = 0; x < grid.x; ++x) {
                g_blockIdx = dim3(x, y, z);
                if (shmem) memset(g_smem.data(), 0xFF, shmem);
                run_block(nthr);
            }
}

}  // namespace hipemu

extern "C" void hipemu_fiber_entry() { hipemu::fiber_main_export(); }
