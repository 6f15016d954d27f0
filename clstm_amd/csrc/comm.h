// comm.h -- the gradient exchange of the data-parallel ranks: the library's communicator (RCCL, bound at first use;
// the host emulator: a shared-memory all-reduce between rank processes), the one-shot peer-read all-reduce over HIP IPC mappings
// (exchange slots, flag handshake, set-up probe, the hosts' announce words) and the replica check's buffers.  Included by
// clstm_hip.hip inside namespace clstm, behind Error / REQUIRE / HIPCHECK / DevBuf / dev_err_words / check_launch and ops.h (the
// device side: k_peer_barrier, k_peer_fill, k_peer_probe, k_peer_allreduce_update).  Split out of clstm_hip.hip in round 5.
// ---- RCCL communicator (data-parallel gradient exchange) ----------------------------------------------
// librccl.so.1 is bound at first use (dlopen): the library loads and runs single-GPU on a box without RCCL,
// and inside a PyTorch process the already-loaded RCCL/HIP runtime pair is reused (same SONAMEs).
#ifndef CLSTM_HIP_EMU
}  // namespace clstm
#include <dlfcn.h>
#include <rccl/rccl.h>
namespace clstm {
struct RcclApi {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  static RcclApi& get() {
    static RcclApi api = [] {
      RcclApi a;
      void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (!h) throw Error(std::string("cannot load librccl.so.1: ") + dlerror());
      auto sym = [&](const char* n) { void* f = dlsym(h, n); if (!f) throw Error(std::string("librccl: missing symbol ") + n); return f; };
      a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
      a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
      a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
      a.AllReduce = (decltype(a.AllReduce))sym("ncclAllReduce");
      a.AllGather = (decltype(a.AllGather))sym("ncclAllGather");
      a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
      return a;
    }();
    return api;
  }
};
#define RCCLCHECK(expr)                                                                         \
  do {                                                                                          \
    ncclResult_t r_ = (expr);                                                                   \
    if (r_ != ncclSuccess) throw Error(std::string(#expr) + " failed: " + RcclApi::get().GetErrorString(r_)); \
  } while (0)
// One-shot peer-read all-reduce (ops.h: k_peer_barrier / k_peer_allreduce_update): every rank owns an exchange buffer of two
// slots (step parity) and a flag array; the other ranks map both through HIP IPC (hipIpcGetMemHandle / hipIpcOpenMemHandle:
// between GPUs the mapping goes over xGMI; two rank processes on ONE device work the same way, which is how a one-GPU box
// tests the protocol).  IPC mappings exist between processes of one host only, so the handles travel through a POSIX
// shared-memory rendezvous named after the communicator's id (no RCCL involved: RCCL refuses two ranks on one device, and
// the test needs exactly that); every rank publishes whether it could export and map, and the peer path is used only if ALL
// ranks could -- otherwise all stay on ncclAllReduce.  Buffers up to PEER_MAX_FLOATS (4 MB): the 35 MB gradient of configs[4]
// is bandwidth-bound and stays with RCCL.
constexpr size_t PEER_MAX_FLOATS = 1u << 20;
struct PeerExchange {
  bool tried = false, ok = false;
  size_t cap = 0;                       // floats per slot
  float* xbuf = nullptr;                // own exchange buffer [2][cap]
  int* flags = nullptr;                 // own flag array [2][PEER_MAX_RANKS]
  float* px[PEER_MAX_RANKS] = {};       // every rank's buffer / flags as mapped here (own rank: the own pointers)
  int* pf[PEER_MAX_RANKS] = {};
  int seq = 0;                          // all-reduces so far (identical on every rank)
  float* slot_ptr(int sq) const { return xbuf + (size_t)(sq & 1) * cap; }
  PeerArgs args(int sq, int rank, int nranks) const {
    PeerArgs a{};
    a.nranks = nranks; a.rank = rank;
    for (int r = 0; r < nranks; r++) { a.x[r] = px[r] + (size_t)(sq & 1) * cap; a.f[r] = pf[r] + (sq & 1) * PEER_MAX_RANKS; }
    return a;
  }
};
}  // namespace clstm
#include <atomic>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <signal.h>
#include <cerrno>
namespace clstm {
// seconds from an environment variable (a hang detector's bound, read once per use site)
static double env_seconds(const char* name, double dflt) {
  const char* e = getenv(name);
  if (!e || !*e) return dflt;
  const double v = atof(e);
  return v > 0 ? v : dflt;
}
// ticks of wall_clock() (100 MHz) k_peer_barrier waits for a peer whose HOST has already announced the step
static long long peer_device_timeout_ticks() { return (long long)(env_seconds("CLSTM_PEER_TIMEOUT_S", 120.0) * 1e8); }
// What the hosts of a communicator share besides the set-up handshake: announced[r] = the last exchange sequence number rank r's
// host is about to enqueue the device barrier for; left[r] = rank r has destroyed its communicator (or failed).
struct PeerHostWords { std::atomic<int> announced[PEER_MAX_RANKS], left[PEER_MAX_RANKS], pid[PEER_MAX_RANKS]; };
// Before a rank enqueues k_peer_barrier for sequence number sq it announces sq and waits -- on the HOST, as long as it takes,
// like ncclAllReduce would -- until every rank's host has announced it too.  A rank whose host is busy elsewhere (clstmocrtrain's
// rank 0 runs the test set and saves while the others are already at the next step) therefore stalls its peers' hosts, not
// their GPUs' watchdog: the device barrier only ever waits for queued device work.  The wait is unbounded, like a collective's;
// a peer that has left the communicator ends it with an error at once.
// A rank that died without running ~Comm (SIGKILL, _exit, a crash) never sets left[r]: about once a second of waiting the
// waiter asks the kernel whether the process that announced as rank r still exists; the wait itself stays unbounded
// for a LIVE peer, like a collective's.
static void peer_liveness(PeerHostWords* w, int r, int sq, int spins) {
  if (spins < 2000 || (spins - 2000) % 10000 != 0) return;
  const int pid = w->pid[r].load();
  if (pid > 0 && kill((pid_t)pid, 0) != 0 && errno == ESRCH)
    throw Error("gradient exchange: the process of rank " + std::to_string(r) + " (pid " + std::to_string(pid) + ") is gone (exchange " + std::to_string(sq) + " never announced)");
}
static void peer_announce_and_wait(PeerHostWords* w, int rank, int nranks, int sq) {
  if (!w) return;
  w->pid[rank].store((int)getpid());
  w->announced[rank].store(sq);
  for (int r = 0; r < nranks; r++) {
    int spins = 0;
    while ((int)((unsigned)sq - (unsigned)w->announced[r].load()) > 0) {   // (wrap-safe: rank r is still behind sq)
      if (w->left[r].load()) throw Error("gradient exchange: rank " + std::to_string(r) + " has left the communicator (exchange " + std::to_string(sq) + " never announced)");
      if (++spins < 2000) sched_yield(); else usleep(100);
      peer_liveness(w, r, sq, spins);
    }
  }
}
struct Comm {
  ncclComm_t comm = nullptr;            // null: a communicator WITHOUT RCCL (CLSTM_COMM_NO_RCCL=1, tests) -- peer path only
  int rank = 0, nranks = 1;
  char id[CLSTM_COMM_ID_BYTES] = {};
  PeerExchange peer;
  DevBuf<float> chk;                    // replica check: [own 4 | summed 4] checksum pieces (ops.h:k_param_checksum)
  DevBuf<unsigned> chk_acc;
  void peer_barrier(int sq, hipStream_t s) {
    peer_announce_and_wait(rv ? &rv->hw : nullptr, rank, nranks, sq);
    const PeerArgs pa = peer.args(sq, rank, nranks);
    CLSTM_LAUNCH(k_peer_barrier, dim3(1), dim3(64), 0, s, pa, sq, dev_err_words() + 6, peer_device_timeout_ticks());
  }
  void allreduce(float* buf, long long n, hipStream_t s) {
    if (comm) { RCCLCHECK(RcclApi::get().AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, comm, s)); return; }
    if (nranks == 1) return;
    // no RCCL: the peer path as a plain in-place all-reduce (copy into the exchange slot, barrier, rank-ordered sum)
    REQUIRE(peer_ready((size_t)n, s), "communicator without RCCL: the ranks could not map each other's exchange buffers (HIP IPC)");
    const int sq = ++peer.seq;
    HIPCHECK(hipMemcpyAsync(peer.slot_ptr(sq), buf, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, s));
    peer_barrier(sq, s);
    const PeerArgs pa = peer.args(sq, rank, nranks);
    CLSTM_LAUNCH(k_peer_allreduce_update, dim3(nblocks((size_t)(n + 3) / 4)), dim3(256), 0, s, pa, (float*)nullptr, (float*)nullptr, buf, (size_t)n, 0.0f, 0.0f, 0.0f,
                 (const int*)nullptr, (int*)nullptr, 0, (int*)nullptr, 0);
    check_launch();
  }
  // rendezvous of the ranks of this host: a shared segment named after the communicator id.  It stays mapped for the life of
  // the communicator (the hosts' announce words live in it); its NAME goes as soon as every rank has it open.
  struct Handles { hipIpcMemHandle_t x, f; int ok; int pad[3]; };
  struct Rendezvous { std::atomic<int> magic, arrived, mapped, good, probed, probe_good; PeerHostWords hw; Handles h[PEER_MAX_RANKS]; };
  Rendezvous* rv = nullptr;
  static bool wait_for(std::atomic<int>& w, int target, double seconds) {
    const auto t0 = std::chrono::steady_clock::now();
    while (w.load() < target) {
      usleep(200);
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) return false;
    }
    return true;
  }
  // The four probe rounds of peer_ready: sequence numbers no training step uses, alternating slots like the steps do
  // (-5: slot 1, -4: slot 0, -3: slot 1, -2: slot 0; the first step is 1: slot 1), every slot written twice with
  // different patterns (ops.h:k_peer_fill / k_peer_probe).
  bool probe_rounds(hipStream_t s) {
    PeerExchange& p = peer;
    int* perr = nullptr;
    if (hipMalloc((void**)&perr, sizeof(int)) != hipSuccess) { (void)hipGetLastError(); return false; }
    bool ran = hipMemsetAsync(perr, 0, sizeof(int), s) == hipSuccess;
    for (int round = 0; ran && round < 4; round++) {
      const int sq = -5 + round;
      const PeerArgs pa = p.args(sq, rank, nranks);
      CLSTM_LAUNCH(k_peer_fill, dim3(nblocks(p.cap)), dim3(256), 0, s, p.slot_ptr(sq), p.cap, rank, round);
      CLSTM_LAUNCH(k_peer_barrier, dim3(1), dim3(64), 0, s, pa, sq, perr, (long long)(20.0 * 1e8));
      CLSTM_LAUNCH(k_peer_probe, dim3(nblocks(p.cap)), dim3(256), 0, s, pa, p.cap, round, perr);
      ran = hipGetLastError() == hipSuccess;
    }
    int e = 1;
    ran = ran && hipMemcpyAsync(&e, perr, sizeof(int), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess && e == 0;
    if (!ran) (void)hipGetLastError();
    // (the slots keep the last patterns: a peer may still be reading them, and every later user assigns what it reads --
    //  the gradient reductions write all nparams elements, the readers' descriptors end there)
    (void)hipFree(perr);
    return ran;
  }
  // collective (every rank calls it at the same point of its first one-call training step): true if the peer path is up
  bool peer_ready(size_t n, hipStream_t s) {
    PeerExchange& p = peer;
    if (p.tried) return p.ok && n <= p.cap;
    p.tried = true;
    static const bool on = !(getenv("CLSTM_PEER_ALLREDUCE") && atoi(getenv("CLSTM_PEER_ALLREDUCE")) == 0);
    if (!on || nranks < 2 || nranks > PEER_MAX_RANKS || n > PEER_MAX_FLOATS) return false;
    HIPCHECK(hipStreamSynchronize(s));
    Handles mine{};
    mine.ok = 1;
    p.cap = PEER_MAX_FLOATS;   // (not the first caller's n: a 4-float replica check may come before the first gradient exchange)
    // Exchange slots AND flags are fine-grained device memory: what a peer reads through its mapping while kernels of the owner
    // are still running must not depend on when the owner's L2 writes a line back, nor on a cache of the reader's side holding
    // the slot's lines of two steps ago -- fine-grained allocations are coherent at system scope by construction (the readers
    // also use system-scope loads, ops.h).  0.5 MB written once per step by the gradient reductions: the uncached stores cost
    // nothing measurable.  Coarse-grained memory only if the fine-grained allocation fails; the probe below judges either.
    if (hipExtMallocWithFlags((void**)&p.xbuf, 2 * p.cap * sizeof(float), hipDeviceMallocFinegrained) != hipSuccess) {
      (void)hipGetLastError();
      if (hipMalloc((void**)&p.xbuf, 2 * p.cap * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); mine.ok = 0; p.xbuf = nullptr; }
    }
    if (hipExtMallocWithFlags((void**)&p.flags, 2 * PEER_MAX_RANKS * sizeof(int), hipDeviceMallocFinegrained) != hipSuccess) {
      (void)hipGetLastError();
      if (hipMalloc((void**)&p.flags, 2 * PEER_MAX_RANKS * sizeof(int)) != hipSuccess) { (void)hipGetLastError(); mine.ok = 0; p.flags = nullptr; }
    }
    if (mine.ok) {
      HIPCHECK(hipMemset(p.xbuf, 0, 2 * p.cap * sizeof(float)));
      HIPCHECK(hipMemset(p.flags, 0, 2 * PEER_MAX_RANKS * sizeof(int)));
      HIPCHECK(hipDeviceSynchronize());
      if (hipIpcGetMemHandle(&mine.x, p.xbuf) != hipSuccess || hipIpcGetMemHandle(&mine.f, p.flags) != hipSuccess) { (void)hipGetLastError(); mine.ok = 0; }
    }
    // the segment: whoever comes first creates it (ranks of another host never arrive here: time-out -> RCCL for everybody)
    unsigned long long hsh = 1469598103934665603ull;
    for (int i = 0; i < CLSTM_COMM_ID_BYTES; i++) hsh = (hsh ^ (unsigned char)id[i]) * 1099511628211ull;
    char name[64];
    snprintf(name, sizeof name, "/clstm_px_%016llx", hsh);
    const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd >= 0 && ftruncate(fd, sizeof(Rendezvous)) == 0) {
      void* m = mmap(nullptr, sizeof(Rendezvous), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
      if (m != MAP_FAILED) rv = (Rendezvous*)m;
    }
    if (fd >= 0) close(fd);
    bool good = rv != nullptr;
    if (rv) {
      const double setup_s = 60.0;
      rv->h[rank] = mine;
      rv->arrived.fetch_add(1);
      good = wait_for(rv->arrived, nranks, setup_s);
      for (int r = 0; good && r < nranks; r++) good = rv->h[r].ok != 0;
      for (int r = 0; good && r < nranks; r++) {
        if (r == rank) { p.px[r] = p.xbuf; p.pf[r] = p.flags; continue; }
        void *x = nullptr, *f = nullptr;
        if (hipIpcOpenMemHandle(&x, rv->h[r].x, hipIpcMemLazyEnablePeerAccess) != hipSuccess ||
            hipIpcOpenMemHandle(&f, rv->h[r].f, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); good = false; }
        p.px[r] = (float*)x; p.pf[r] = (int*)f;
      }
      // ... and they agree: the peer path only if EVERY rank mapped every other rank
      if (good) rv->good.fetch_add(1);
      rv->mapped.fetch_add(1);
      const bool all_here = wait_for(rv->mapped, nranks, setup_s);
      good = all_here && rv->good.load() == nranks;
      // ... and the mappings DELIVER: four handshakes through the mapped flag arrays, each followed by a look at EVERY element of
      // every rank's slot through this rank's mapping of it (probe_rounds) -- a path that maps but does not deliver (flags that
      // never arrive, stale, partial or foreign data) falls back to RCCL here instead of failing at the first step
      if (good) {
        if (probe_rounds(s)) rv->probe_good.fetch_add(1);
        rv->probed.fetch_add(1);
        good = wait_for(rv->probed, nranks, 2 * setup_s) && rv->probe_good.load() == nranks;
      }
      if (rank == 0) shm_unlink(name);          // (every rank that will ever come has it open or has given up)
    }
    p.ok = good;
    if (!p.ok) peer_release();
    return p.ok && n <= p.cap;
  }
  void peer_release() {
    PeerExchange& p = peer;
    if (rv) { rv->hw.left[rank].store(1); munmap(rv, sizeof(Rendezvous)); rv = nullptr; }
    for (int r = 0; r < PEER_MAX_RANKS; r++) {
      if (r != rank && p.px[r]) (void)hipIpcCloseMemHandle(p.px[r]);
      if (r != rank && p.pf[r]) (void)hipIpcCloseMemHandle(p.pf[r]);
      p.px[r] = nullptr; p.pf[r] = nullptr;
    }
    if (p.xbuf) (void)hipFree(p.xbuf);
    if (p.flags) (void)hipFree(p.flags);
    p.xbuf = nullptr; p.flags = nullptr; p.ok = false;
  }
  ~Comm() {
    peer_release();
    chk.release(); chk_acc.release();
    if (comm) (void)RcclApi::get().CommDestroy(comm);
  }
};
#else
}  // namespace clstm
#include <atomic>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <signal.h>
#include <cerrno>
namespace clstm {
// Host emulator: the ranks are host PROCESSES and the communicator is a POSIX shared-memory segment (one slot of
// SLOT floats per rank + a sense-reversing barrier) -- so that the world-size-2 CPU test drives the same entry points
// and the same in-library order (all-reduce of g -> d += g -> update) as the RCCL build.  Every rank sums the slots in
// rank order: bit-identical results on all ranks, like a deterministic all-reduce.
constexpr size_t PEER_MAX_FLOATS = 1u << 18;
struct PeerExchange {   // (host emulator: the "mapped" buffers and flags of the ranks are regions of the shared segment)
  bool tried = false, ok = false;
  size_t cap = 0;                          // floats of a slot in use (the slots lie PEER_MAX_FLOATS apart)
  float* xbuf = nullptr;
  int* flags = nullptr;
  float* px[PEER_MAX_RANKS] = {};
  int* pf[PEER_MAX_RANKS] = {};
  int seq = 0;
  float* slot_ptr(int sq) const { return xbuf + (size_t)(sq & 1) * PEER_MAX_FLOATS; }
  PeerArgs args(int sq, int rank, int nranks) const {
    PeerArgs a{};
    a.nranks = nranks; a.rank = rank;
    for (int r = 0; r < nranks; r++) { a.x[r] = px[r] + (size_t)(sq & 1) * PEER_MAX_FLOATS; a.f[r] = pf[r] + (sq & 1) * PEER_MAX_RANKS; }
    return a;
  }
};
// (the hosts' announce words: see the GPU build's PeerHostWords / peer_announce_and_wait above -- same protocol)
struct PeerHostWords { std::atomic<int> announced[PEER_MAX_RANKS], left[PEER_MAX_RANKS], pid[PEER_MAX_RANKS]; };
static double env_seconds(const char* name, double dflt) {
  const char* e = getenv(name);
  if (!e || !*e) return dflt;
  const double v = atof(e);
  return v > 0 ? v : dflt;
}
static long long peer_device_timeout_ticks() { return (long long)(env_seconds("CLSTM_PEER_TIMEOUT_S", 120.0) * 1e8); }
// A rank that died without running ~Comm (SIGKILL, _exit, a crash) never sets left[r]: about once a second of waiting the
// waiter asks the kernel whether the process that announced as rank r still exists; the wait itself stays unbounded
// for a LIVE peer, like a collective's.
static void peer_liveness(PeerHostWords* w, int r, int sq, int spins) {
  if (spins < 2000 || (spins - 2000) % 10000 != 0) return;
  const int pid = w->pid[r].load();
  if (pid > 0 && kill((pid_t)pid, 0) != 0 && errno == ESRCH)
    throw Error("gradient exchange: the process of rank " + std::to_string(r) + " (pid " + std::to_string(pid) + ") is gone (exchange " + std::to_string(sq) + " never announced)");
}
static void peer_announce_and_wait(PeerHostWords* w, int rank, int nranks, int sq) {
  if (!w) return;
  w->pid[rank].store((int)getpid());
  w->announced[rank].store(sq);
  for (int r = 0; r < nranks; r++) {
    int spins = 0;
    while ((int)((unsigned)sq - (unsigned)w->announced[r].load()) > 0) {
      if (w->left[r].load()) throw Error("gradient exchange: rank " + std::to_string(r) + " has left the communicator (exchange " + std::to_string(sq) + " never announced)");
      if (++spins < 2000) sched_yield(); else usleep(100);
      peer_liveness(w, r, sq, spins);
    }
  }
}
struct Comm {
  static const long long SLOT = 1 << 18;
  struct Shm { std::atomic<int> magic, arrived, gen; int pad; PeerHostWords hw; float slots[1]; };
  int rank = 0, nranks = 1;
  PeerExchange peer;
  DevBuf<float> chk;
  DevBuf<unsigned> chk_acc;
  void peer_barrier(int sq, hipStream_t s) {
    peer_announce_and_wait(shm ? &shm->hw : nullptr, rank, nranks, sq);
    const PeerArgs pa = peer.args(sq, rank, nranks);
    CLSTM_LAUNCH(k_peer_barrier, dim3(1), dim3(64), 0, s, pa, sq, dev_err_words() + 6, peer_device_timeout_ticks());
  }
  // per rank behind the all-reduce slots: exchange buffer [2][SLOT] floats, then flags [2][PEER_MAX_RANKS] ints (zero pages)
  static size_t peer_region_bytes() { return (size_t)2 * SLOT * sizeof(float) + 2 * PEER_MAX_RANKS * sizeof(int) + 64; }
  bool peer_ready(size_t n, hipStream_t s) {
    PeerExchange& p = peer;
    if (p.tried) return p.ok && n <= p.cap;
    p.tried = true;
    const bool on = !(getenv("CLSTM_PEER_ALLREDUCE") && atoi(getenv("CLSTM_PEER_ALLREDUCE")) == 0);
    if (!on || nranks < 2 || nranks > PEER_MAX_RANKS || n > (size_t)SLOT || !shm) return false;
    char* base = (char*)shm + sizeof(Shm) + (size_t)nranks * SLOT * sizeof(float);
    for (int r = 0; r < nranks; r++) {
      p.px[r] = (float*)(base + (size_t)r * peer_region_bytes());
      p.pf[r] = (int*)(base + (size_t)r * peer_region_bytes() + (size_t)2 * SLOT * sizeof(float));
    }
    p.cap = PEER_MAX_FLOATS; p.xbuf = p.px[rank]; p.flags = p.pf[rank];   // (as the GPU build: never the first caller's n)
    // the set-up probe of the GPU build, same kernels (ops.h:k_peer_fill / k_peer_probe), over the slot length in use
    int perr = 0;
    for (int round = 0; round < 4; round++) {
      const int sq = -5 + round;
      const PeerArgs pa = p.args(sq, rank, nranks);
      CLSTM_LAUNCH(k_peer_fill, dim3(nblocks(p.cap)), dim3(256), 0, s, p.slot_ptr(sq), p.cap, rank, round);
      CLSTM_LAUNCH(k_peer_barrier, dim3(1), dim3(64), 0, s, pa, sq, &perr, (long long)(60.0 * 1e8));
      CLSTM_LAUNCH(k_peer_probe, dim3(nblocks(p.cap)), dim3(256), 0, s, pa, p.cap, round, &perr);
    }
    // (the slots keep the last patterns: a peer may still be reading them; every later user assigns what it reads)
    REQUIRE(perr == 0, "emulator communicator: the peer probe failed");
    p.ok = true;
    return true;
  }
  Shm* shm = nullptr;
  size_t bytes = 0;
  std::string name;
  void open(const char* id, int rank_, int nranks_) {
    rank = rank_; nranks = nranks_;
    if (nranks == 1) return;
    name.assign(id, strnlen(id, CLSTM_COMM_ID_BYTES));
    REQUIRE(!name.empty() && name[0] == '/', "emulator communicator: bad id");
    bytes = sizeof(Shm) + (size_t)nranks * SLOT * sizeof(float) + (size_t)nranks * peer_region_bytes();
    int fd = -1;
    if (rank == 0) {
      fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
      REQUIRE(fd >= 0 && ftruncate(fd, (off_t)bytes) == 0, "emulator communicator: cannot create the shared segment");
    } else {
      for (int tries = 0; tries < 20000 && fd < 0; tries++) { fd = shm_open(name.c_str(), O_RDWR, 0600); if (fd < 0) usleep(1000); }
      REQUIRE(fd >= 0, "emulator communicator: rank 0's segment did not appear");
      struct stat st;
      for (int tries = 0; tries < 20000; tries++) { if (fstat(fd, &st) == 0 && (size_t)st.st_size >= bytes) break; usleep(1000); }
    }
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    REQUIRE(p != MAP_FAILED, "emulator communicator: mmap failed");
    shm = (Shm*)p;
    if (rank == 0) { shm->arrived.store(0); shm->gen.store(0); shm->magic.store(0x434c5354); }
    else for (int tries = 0; shm->magic.load() != 0x434c5354; tries++) { REQUIRE(tries < 20000, "emulator communicator: rank 0 never initialised the segment"); usleep(1000); }
    barrier();
    if (rank == 0) shm_unlink(name.c_str());   // every rank has it mapped: the name can go
  }
  void barrier() {
    const int g = shm->gen.load();
    if (shm->arrived.fetch_add(1) + 1 == nranks) { shm->arrived.store(0); shm->gen.store(g + 1); }
    else while (shm->gen.load() == g) sched_yield();
  }
  void allreduce(float* buf, long long n, hipStream_t) {
    if (nranks == 1) return;
    for (long long o = 0; o < n; o += SLOT) {
      const long long m = std::min(SLOT, n - o);
      memcpy(shm->slots + (size_t)rank * SLOT, buf + o, (size_t)m * sizeof(float));
      barrier();
      for (long long i = 0; i < m; i++) {
        float acc = shm->slots[i];
        for (int r = 1; r < nranks; r++) acc += shm->slots[(size_t)r * SLOT + i];
        buf[o + i] = acc;
      }
      barrier();
    }
  }
  ~Comm() { if (shm) { shm->hw.left[rank].store(1); munmap(shm, bytes); } chk.release(); chk_acc.release(); }
};
#endif

