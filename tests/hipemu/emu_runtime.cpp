// Fiber scheduler of the wave64 emulator (test infrastructure, see hip/hip_runtime.h).
#include "hip/hip_runtime.h"
#include <setjmp.h>
#include <ucontext.h>
#include <deque>
#include <pthread.h>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

thread_local unsigned char emu_dyn_smem[160 * 1024];
thread_local emu_uint3 threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;

namespace {
constexpr size_t kStack = 96 * 1024;

// Fibers live as long as their worker thread and are switched with _setjmp / _longjmp (round 5): getcontext / swapcontext
// save and restore the signal mask -- a system call per switch, several per fiber and block; only a fiber's FIRST entry onto
// its stack still goes through swapcontext.  A fiber runs the kernel of every block its worker executes, one after the other.
struct Fiber {
  ucontext_t ctx;
  jmp_buf jb;
  char* stack = nullptr;
  bool started = false;
  bool done = true;
  const volatile unsigned* wait_ptr = nullptr;  // runnable when *wait_ptr != wait_val
  unsigned wait_val = 0;
  emu_uint3 tid;
};

struct Wave {
  volatile unsigned gen = 0;
  int arrived = 0;
  int nlanes = 64;
  unsigned ops = 0;
  alignas(16) char buf[2][64][64];
};

struct Worker {
  std::deque<Fiber> fibers;  // (stable addresses: a ucontext_t must not move once made)
  std::vector<Wave> waves;
  ucontext_t sched;
  jmp_buf sched_jb;
  int cur = -1;
  int nthreads = 0;
  int active = 0;
  volatile unsigned bar_gen = 0;
  int bar_arrived = 0;
  void (*tramp)(void*) = nullptr;
  void* args = nullptr;
  // launches run on short-lived threads, each with its own thread_local Worker: the fiber stacks must go with it
  ~Worker() {
    for (Fiber& f : fibers) free(f.stack);
  }
};
thread_local Worker* W = nullptr;

void fiber_main() {
  Worker* w = W;  // (a fiber stays with the thread that made it)
  for (;;) {      // one kernel invocation per block this fiber takes part in
    w->tramp(w->args);
    Fiber& f = w->fibers[w->cur];
    f.done = true;
    w->active--;
    // a thread that exits counts as arrived for any pending workgroup barrier
    if (w->bar_arrived > 0 && w->bar_arrived >= w->active) { w->bar_arrived = 0; w->bar_gen++; }
    if (!_setjmp(f.jb)) _longjmp(w->sched_jb, 1);
  }
}

void yield_wait(const volatile unsigned* p, unsigned v) {
  Worker* w = W;
  Fiber& f = w->fibers[w->cur];
  f.wait_ptr = p;
  f.wait_val = v;
  if (!_setjmp(f.jb)) _longjmp(w->sched_jb, 1);
}

void run_block(Worker* w, dim3 block) {
  int n = block.x * block.y * block.z;
  while ((int)w->fibers.size() < n) {
    w->fibers.emplace_back();
    Fiber& f = w->fibers.back();
    f.stack = (char*)malloc(kStack);
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack;
    f.ctx.uc_stack.ss_size = kStack;
    f.ctx.uc_link = nullptr;
    makecontext(&f.ctx, fiber_main, 0);
  }
  int nw = (n + 63) / 64;
  w->waves.assign(nw, Wave());
  for (int i = 0; i < nw; ++i) w->waves[i].nlanes = std::min(64, n - 64 * i);
  w->nthreads = n;
  w->active = n;
  w->bar_gen = 0;
  w->bar_arrived = 0;
  for (int i = 0; i < n; ++i) {
    Fiber& f = w->fibers[i];
    f.done = false;
    f.wait_ptr = nullptr;
    f.tid.x = i % block.x;
    f.tid.y = (i / block.x) % block.y;
    f.tid.z = i / (block.x * block.y);
  }
  int remaining = n;
  while (remaining > 0) {
    bool progressed = false;
    remaining = 0;
    for (int i = 0; i < n; ++i) {
      Fiber& f = w->fibers[i];
      if (f.done) continue;
      remaining++;
      if (f.wait_ptr && *f.wait_ptr == f.wait_val) continue;
      f.wait_ptr = nullptr;
      w->cur = i;
      threadIdx = f.tid;
      if (!_setjmp(w->sched_jb)) {
        if (!f.started) {
          f.started = true;
          swapcontext(&w->sched, &f.ctx);  // (first entry onto the fiber's stack; it comes back through sched_jb)
        } else {
          _longjmp(f.jb, 1);
        }
      }
      progressed = true;
    }
    if (!progressed && remaining > 0) {
      fprintf(stderr, "hipemu: deadlock (divergent barrier / collective) in block (%u,%u,%u)\n",
              blockIdx.x, blockIdx.y, blockIdx.z);
      abort();
    }
  }
}
}  // namespace

int emu_lane() { return W->cur & 63; }

void emu_syncthreads() {
  Worker* w = W;
  unsigned g = w->bar_gen;
  w->bar_arrived++;
  if (w->bar_arrived >= w->active) {
    w->bar_arrived = 0;
    w->bar_gen = g + 1;
    return;
  }
  yield_wait(&w->bar_gen, g);
}

void emu_wave_exchange(const void* in, void* all_out, int bytes) {
  Worker* w = W;
  if (bytes > 64) { fprintf(stderr, "hipemu: exchange too wide\n"); abort(); }
  Wave& wv = w->waves[w->cur >> 6];
  int lane = w->cur & 63;
  // every lane runs the same op sequence; its own count picks the double buffer
  static thread_local std::vector<unsigned> lane_ops;
  if (lane_ops.size() < (size_t)w->nthreads) lane_ops.resize(w->nthreads);
  (void)lane_ops;
  unsigned g = wv.gen;
  int b = g & 1;
  memcpy(wv.buf[b][lane], in, bytes);
  wv.arrived++;
  if (wv.arrived >= wv.nlanes) {
    wv.arrived = 0;
    wv.gen = g + 1;
  } else {
    yield_wait(&wv.gen, g);
  }
  char* o = (char*)all_out;
  for (int i = 0; i < 64; ++i) memcpy(o + (size_t)i * bytes, wv.buf[b][i], bytes);
}

// Worker threads that live as long as the process (round 5): a launch used to START up to 8 threads, each of which
// allocated -- and at its exit freed -- the fiber stacks of its thread_local Worker; with thousands of launches per test the
// CPU suite spent a quarter of its time in the kernel (13 of 57 CPU-minutes).  The pool's threads keep their Workers; one
// launch runs at a time (launches are synchronous anyway; a second host thread waits its turn); after a fork() the child
// builds a pool of its own.
namespace {
struct Pool {
  std::mutex launch_mu;
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  std::function<void()> job;
  unsigned size = 0, want = 0, finished = 0;
  unsigned long gen = 0;
};
Pool* g_pool = nullptr;
std::once_flag g_atfork;

void pool_thread(Pool* p, unsigned idx) {
  unsigned long seen = 0;
  for (;;) {
    std::function<void()> job;
    {
      std::unique_lock<std::mutex> lk(p->mu);
      p->cv_work.wait(lk, [&] { return p->gen != seen; });
      seen = p->gen;
      if (idx >= p->want) continue;
      job = p->job;
    }
    job();
    {
      std::lock_guard<std::mutex> lk(p->mu);
      if (++p->finished == p->want) p->cv_done.notify_all();
    }
  }
}

Pool* get_pool() {
  std::call_once(g_atfork, [] { pthread_atfork(nullptr, nullptr, [] { g_pool = nullptr; }); });
  if (!g_pool) {
    Pool* p = new Pool();  // (never destroyed: its threads are detached and end with the process)
    unsigned hw = std::thread::hardware_concurrency();
    p->size = hw ? hw : 4;
    if (p->size < 8) p->size = 8;  // (HIPEMU_THREADS may ask for up to this many)
    for (unsigned i = 0; i < p->size; ++i) std::thread(pool_thread, p, i).detach();
    g_pool = p;
  }
  return g_pool;
}
}  // namespace

void emu_launch_impl(void (*tramp)(void*), void* args, dim3 grid, dim3 block) {
  size_t nblocks = (size_t)grid.x * grid.y * grid.z;
  if (nblocks == 0) return;
  unsigned hw = std::thread::hardware_concurrency();
  const char* env = getenv("HIPEMU_THREADS");
  unsigned nthr = env ? (unsigned)atoi(env) : (hw ? hw : 4);
  nthr = (unsigned)std::min<size_t>(nthr, nblocks);
  std::atomic<size_t> next{0};
  auto work = [&]() {
    static thread_local Worker worker;
    W = &worker;
    worker.tramp = tramp;
    worker.args = args;
    blockDim = block;
    gridDim = grid;
    for (;;) {
      size_t b = next.fetch_add(1);
      if (b >= nblocks) break;
      blockIdx.x = b % grid.x;
      blockIdx.y = (b / grid.x) % grid.y;
      blockIdx.z = b / ((size_t)grid.x * grid.y);
      run_block(&worker, block);
    }
  };
  if (nthr <= 1) { work(); return; }
  Pool* p = get_pool();
  if (nthr > p->size) nthr = p->size;
  std::lock_guard<std::mutex> one(p->launch_mu);
  {
    std::lock_guard<std::mutex> lk(p->mu);
    p->job = work;
    p->want = nthr;
    p->finished = 0;
    ++p->gen;
  }
  p->cv_work.notify_all();
  std::unique_lock<std::mutex> lk(p->mu);
  p->cv_done.wait(lk, [&] { return p->finished == p->want; });
}
