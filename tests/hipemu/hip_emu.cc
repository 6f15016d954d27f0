// Runtime of the test-only emulator (see hip_emu.h): thread-local "hardware registers", the fiber switch and
// the workgroup scheduler.
#include "hip_emu.h"
#include <sys/mman.h>
#include <atomic>
#if !defined(__x86_64__)
#include <ucontext.h>
#endif

thread_local uint3_emu threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;
thread_local EmuBlock* emu_blk = nullptr;

namespace {
constexpr size_t STACK_BYTES = 128 << 10;

// per-OS-thread pool of fiber stacks (mmap'd once, reused by every workgroup this thread runs)
struct StackPool {
  std::vector<char*> stacks;
  char* get(size_t i) {
    while (stacks.size() <= i) {
      void* p = mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
      if (p == MAP_FAILED) { perror("emu: mmap fiber stack"); abort(); }
      stacks.push_back((char*)p);
    }
    return stacks[i];
  }
  ~StackPool() { for (char* p : stacks) munmap(p, STACK_BYTES); }
};
thread_local StackPool pool;

#if defined(__x86_64__)
// emu_ctx_switch(void** save_sp, void* load_sp): callee-saved registers on the stack, swap rsp
extern "C" void emu_ctx_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl emu_ctx_switch
.type emu_ctx_switch,@function
emu_ctx_switch:
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
.size emu_ctx_switch,.-emu_ctx_switch
)");
void fiber_entry();
void* make_context(char* stack) {
  uintptr_t top = ((uintptr_t)stack + STACK_BYTES) & ~(uintptr_t)15;
  void** sp = (void**)top;
  *--sp = nullptr;                 // return-address slot of fiber_entry (never returns)
  *--sp = (void*)&fiber_entry;     // popped by `ret`
  for (int i = 0; i < 6; i++) *--sp = nullptr;
  return sp;
}
inline void switch_to(void** save, void* load) { emu_ctx_switch(save, load); }
#else
struct UCtx { ucontext_t c; };
void fiber_entry();
void* make_context(char* stack) {
  UCtx* u = new UCtx;
  getcontext(&u->c);
  u->c.uc_stack.ss_sp = stack;
  u->c.uc_stack.ss_size = STACK_BYTES;
  u->c.uc_link = nullptr;
  makecontext(&u->c, fiber_entry, 0);
  return u;
}
thread_local UCtx main_ctx;
inline void switch_to(void** save, void* load) {
  // *save already points at the UCtx of the running fiber (or is the main slot)
  UCtx* from = *save ? (UCtx*)*save : &main_ctx;
  if (!*save) *save = from;
  swapcontext(&from->c, &((UCtx*)load)->c);
}
#endif

inline void enter(EmuBlock& b, unsigned i) {
  b.cur = i;
  threadIdx = b.fib[i].tid;
}

void retire(EmuBar& bar) {
  if (bar.n) bar.n--;
  if (bar.n && bar.count >= bar.n) { bar.count = 0; bar.gen++; }
}

void fiber_entry() {
  EmuBlock& b = *emu_blk;
  b.body(b.body_arg);
  EmuFiber& f = b.fib[b.cur];
  f.done = true;
  b.live--;
  retire(b.block_bar);               // a wave that has exited no longer takes part in barriers
  retire(b.wave_bar[f.wave]);
  if (b.live == 0) {
    switch_to(&f.sp, b.main_sp);
  } else {
    unsigned i = b.cur;
    do { i = i + 1 == b.fib.size() ? 0 : i + 1; } while (b.fib[i].done);
    void** save = &f.sp;
    enter(b, i);
    switch_to(save, b.fib[i].sp);
  }
  abort();                           // a finished fiber is never resumed
}
}  // namespace

void emu_yield() {
  EmuBlock& b = *emu_blk;
  const unsigned from = b.cur;
  unsigned i = from;
  do { i = i + 1 == b.fib.size() ? 0 : i + 1; } while (b.fib[i].done);
  if (i == from) return;             // alone: the caller is polling something another workgroup owns
  enter(b, i);
  switch_to(&b.fib[from].sp, b.fib[i].sp);
}

void emu_run_block(EmuBlock& blk, dim3 block, void (*body)(void*), void* arg) {
  const unsigned n = block.x * block.y * block.z;
  blk.block_bar = EmuBar{n, 0, 0};
  blk.wave_bar.assign(n / 64, EmuBar{64, 0, 0});
  blk.fib.resize(n);
  for (unsigned t = 0; t < n; t++) {
    blk.fib[t].tid = {t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
    blk.fib[t].wave = t / 64;
    blk.fib[t].done = false;
    blk.fib[t].sp = make_context(pool.get(t));
  }
  blk.live = n;
  blk.body = body;
  blk.body_arg = arg;
  EmuBlock* outer = emu_blk;
  emu_blk = &blk;
  enter(blk, 0);
  blk.main_sp = nullptr;
  switch_to(&blk.main_sp, blk.fib[0].sp);
#if !defined(__x86_64__)
  for (auto& f : blk.fib) delete (UCtx*)f.sp;
#endif
  emu_blk = outer;
}

void emu_parallel_for(unsigned n, void (*fn)(unsigned, void*), void* arg) {
  static const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  const unsigned nt = std::min(n, hw);
  if (nt <= 1) {
    for (unsigned i = 0; i < n; i++) fn(i, arg);
    return;
  }
  std::atomic<unsigned> next{0};
  std::vector<std::thread> th;
  th.reserve(nt);
  for (unsigned t = 0; t < nt; t++)
    th.emplace_back([&]() {
      for (unsigned i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i, arg);
    });
  for (auto& t : th) t.join();
}
