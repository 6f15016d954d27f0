// clstmocrtrain -- the reference's OCR training driver (clstmocrtrain.cc:97-224) on the MI355X path:
// same arguments, environment variables, stdout lines and model files; one text line per update.
// Beyond the reference: batch=N trains on minibatches of N lines per update (the batched device path; a helper
// thread reads and normalises the next minibatch while the GPU works on the current one), nhidden2=M builds the
// "bidi2" prefab.  With the defaults (batch=1, nhidden2=0) the run is the reference's, update for update.
#include "clstmhl.h"
#include <signal.h>
#include <sys/wait.h>
#include <unistd.h>
using namespace clstmhost;

// ---- ngpu=N: one process per GPU ---------------------------------------------------------------------------------
// The parent forks N-1 rank processes BEFORE anything touches the device; rank r binds GPU r (the r-th entry of
// HIP_VISIBLE_DEVICES if the caller set one), rank 0 creates the RCCL id and hands it to the others through pipes, every
// rank joins the library communicator (clstm_comm_create) and attaches it to its network: update() then all-reduces the
// fresh minibatch gradient before the identical parameter update (precedent: share_deltas, clstm.cc:731-744).  All
// ranks draw the SAME minibatches (same lrand48 sequence) and train on their share of each (dealt longest line first, see deal()); only rank 0
// reports, tests and saves.  With ngpu=N batch=B the run equals ngpu=1 batch=B up to the summation order of the gradient.
struct Ranks;
static Ranks* g_ranks = nullptr;
// rank 0: a rank process that dies leaves the others waiting in the next all-reduce for ever -- end the run instead
// (only the recorded rank pids are reaped -- never another child of the process)
static volatile pid_t g_rank_pids[256];
static volatile int g_nrank_pids = 0;
static void on_rank_exit(int) {
  for (int i = 0; i < g_nrank_pids; i++) {
    int st = 0;
    const pid_t k = g_rank_pids[i];
    if (k <= 0 || waitpid(k, &st, WNOHANG) != k) continue;
    g_rank_pids[i] = 0;   // collected
    if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) {
      static const char msg[] = "FATAL: a rank process of ngpu=N ended abnormally\n";
      if (write(2, msg, sizeof msg - 1) < 0) {}
      _exit(1);
    }
  }
}
struct Ranks {
  int rank = 0, n = 1;
  std::vector<pid_t> kids;
  clstm_comm* comm = nullptr;
  void start(int ngpu) {
    n = ngpu;
    if (n <= 1) return;
    std::vector<int> wr(n, -1);
    int rd = -1;
    if (n > 256) fail("ngpu too large");
    // SIGCHLD is blocked while the ranks are forked and recorded and the handler is installed BEFORE the first fork: a rank
    // that dies at once (a bad HIP_VISIBLE_DEVICES entry, an early fail()) is seen when the signal is unblocked below --
    // installed after the loop, the handler missed it and rank 0 waited for ever in clstm_comm_create
    sigset_t chld, old_mask;
    sigemptyset(&chld); sigaddset(&chld, SIGCHLD);
    sigprocmask(SIG_BLOCK, &chld, &old_mask);
    {
      struct sigaction sa{};
      sa.sa_handler = on_rank_exit;
      sa.sa_flags = SA_RESTART | SA_NOCLDSTOP;
      sigaction(SIGCHLD, &sa, nullptr);
    }
    for (int r = 1; r < n; r++) {
      int fd[2];
      if (pipe(fd) != 0) fail("pipe() failed");
      const pid_t pid = fork();
      if (pid < 0) fail("fork() failed");
      if (pid == 0) {   // rank r
        rank = r; rd = fd[0]; close(fd[1]);
        for (int q = 1; q < r; q++) close(wr[q]);
        kids.clear();
        break;
      }
      kids.push_back(pid); wr[r] = fd[1]; close(fd[0]);
      g_rank_pids[g_nrank_pids] = pid; g_nrank_pids = g_nrank_pids + 1;
    }
    if (rank != 0) { g_nrank_pids = 0; signal(SIGCHLD, SIG_DFL); }
    sigprocmask(SIG_SETMASK, &old_mask, nullptr);   // (rank 0: a pending SIGCHLD is delivered here)
    {   // one GPU per rank
      const char* vis = getenv("HIP_VISIBLE_DEVICES");
      std::string dev = std::to_string(rank);
      if (vis && *vis && !getenv("CLSTM_NGPU_SHARE_DEVICE")) {
        std::vector<std::string> ids;
        std::string cur;
        for (const char* c = vis;; c++) { if (*c == ',' || !*c) { ids.push_back(cur); cur.clear(); if (!*c) break; } else cur.push_back(*c); }
        if ((int)ids.size() < n) fail("ngpu exceeds the GPUs listed in HIP_VISIBLE_DEVICES");
        dev = ids[rank];
      }
      if (!getenv("CLSTM_NGPU_SHARE_DEVICE")) setenv("HIP_VISIBLE_DEVICES", dev.c_str(), 1);   // (tests without N GPUs: ranks share the device of the host emulator)
    }
    char id[CLSTM_COMM_ID_BYTES];
    if (rank == 0) {
      chk(clstm_comm_unique_id(id), "clstm_comm_unique_id");
      for (int r = 1; r < n; r++) { if (write(wr[r], id, sizeof id) != (ssize_t)sizeof id) fail("rank pipe write failed"); close(wr[r]); }
    } else {
      size_t got = 0;
      while (got < sizeof id) { const ssize_t k = read(rd, id + got, sizeof id - got); if (k <= 0) fail("rank pipe read failed"); got += (size_t)k; }
      close(rd);
    }
    chk(clstm_comm_create(&comm, id, rank, n), "clstm_comm_create");
  }
  // rank 0 waits for the others; a rank process leaves through here
  int finish(int status) {
    // the final synchronisation is where a device error enqueued since the last read-back surfaces (skipped updates, the NaN
    // flag, a peer time-out, diverged replicas): a rank that saw one must not exit 0 -- rank 0 would report and save as if the
    // replicas agreed
    if (comm && clstm_synchronize() != 0) {
      std::cerr << "FATAL: rank " << rank << ": " << clstm_last_error() << std::endl;
      status = status ? status : 1;
    }
    if (rank != 0) { fflush(nullptr); _exit(status); }
    signal(SIGCHLD, SIG_DFL);   // from here on the exits are collected below
    for (pid_t k : kids) {
      int st = 0;
      const pid_t r = waitpid(k, &st, 0);
      if (r < 0) continue;        // (already collected by the handler: it exited cleanly, or the handler would have ended the run)
      if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) status = status ? status : 1;
    }
    return status;
  }
};

struct Dataset {  // clstmocrtrain.cc:56-76
  vector<string> fnames;
  ustring charsep = utf8_to_utf32(getsenv("charsep", ""));
  int size() { return (int)fnames.size(); }
  void readFileList(const string& file_list) { read_lines(fnames, file_list); }
  void getCodec(Codec& codec) {
    vector<string> gtnames;
    for (auto& s : fnames) gtnames.push_back(basename_noext(s) + ".gt.txt");
    codec.build(gtnames, charsep);
  }
  ustring separate_chars(const ustring& s) {
    if (charsep.empty()) return s;
    ustring r;
    for (size_t i = 0; i < s.size(); i++) {
      if (i > 0) r.push_back(charsep[0]);
      r.push_back(s[i]);
    }
    return r;
  }
  void readSample(Image& raw, ustring& gt, int index) {
    const string& fname = fnames[index];
    gt = separate_chars(utf8_to_utf32(read_text(basename_noext(fname) + ".gt.txt")));
    read_png(raw, fname);
    for (float& v : raw.d) v = -v + 1.0f;  // raw = 1 - raw: ink = 1 (clstmocrtrain.cc:73)
  }
};

static int print_usage(char** argv) {
  std::cerr << "Usage: [VAR=VAL...] " << argv[0] << " TRAININGLIST [TESTLIST]\n"
            << "  Variables: load save_name nhidden lrate momentum target_height ntrain start charsep\n"
            << "             report_time test_every report_every save_every params   (clstmocrtrain.cc:99-115)\n"
            << "             batch (lines per update, default 1)  nhidden2 (> 0: bidi2)  ngpu (processes, one GPU each; needs batch % ngpu == 0)   (not in the reference)\n";
  return EXIT_FAILURE;
}

static int main1(int argc, char** argv) {
  if (argc < 2 || argc > 3 || !strcmp(argv[1], "-h") || !strcmp(argv[1], "--help")) return print_usage(argv);
  int ntrain = getienv("ntrain", 10000000);
  string save_name = getsenv("save_name", "_ocr");
  int report_time = getienv("report_time", 0);
  Dataset trainingset, testset;
  trainingset.readFileList(argv[1]);
  if (trainingset.size() <= 0) fail("empty training list");
  if (argc > 2) testset.readFileList(argv[2]);
  const int batch = std::max(1, getienv("batch", 1));
  const int ngpu = std::max(1, getienv("ngpu", 1));
  if (ngpu > 1 && batch % ngpu != 0) fail("ngpu=N needs batch to be a multiple of N (every rank trains on batch / N lines per update)");
  Ranks ranks;
  g_ranks = &ranks;
  ranks.start(ngpu);            // (forks: nothing above touched the device)
  const bool lead = ranks.rank == 0;
  std::ostream null_out(nullptr);
  std::ostream& out = lead ? std::cout : null_out;   // only rank 0 talks
  out << "got " << trainingset.size() << " files, " << testset.size() << " tests" << std::endl;
  if (ngpu > 1) out << "ranks " << ngpu << " x " << batch / ngpu << " lines" << std::endl;
  string load_name = getsenv("load", "");
  CLSTMOCR clstm;
  if (load_name != "") {
    clstm.load(load_name);
  } else {
    Codec codec;
    trainingset.getCodec(codec);
    out << "got " << codec.size() << " classes" << std::endl;
    clstm.target_height = int(getrenv("target_height", 48));
    const int nhidden2 = getienv("nhidden2", 0);
    if (nhidden2 > 0) clstm.createBidi2(codec.codec, getienv("nhidden", 100), nhidden2);
    else clstm.createBidi(codec.codec, getienv("nhidden", 100));
    clstm.setLearningRate(getdenv("lrate", 1e-4), getdenv("momentum", 0.9));
  }
  if (ranks.comm) chk(clstm_net_set_comm(clstm.net, ranks.comm), "clstm_net_set_comm");
  double test_error = 9999.0, best_error = 1e38;
  double start_time = now();
  int start = atoi(clstm.attr_get("trial", std::to_string(getienv("start", -1))).c_str()) + 1;
  if (start > 0) out << "start " << start << std::endl;
  Trigger test_trigger(getienv("test_every", 10000), -1, start);
  test_trigger.skip0();
  Trigger save_trigger(getienv("save_every", 10000), ntrain, start);
  save_trigger.enable(save_name != "" && lead).skip0();
  Trigger report_trigger(getienv("report_every", 100), ntrain, start);
  // ngpu > 1: the lines of a minibatch are dealt to the ranks LONGEST FIRST, round-robin, so that every rank's longest line
  // (= the length of its recurrence and CTC launches: one workgroup per line and direction) is the same to within one
  // position of the sorted order -- with contiguous shards the all-reduce waits for whichever rank drew the longest lines
  // (ragged T ~ U{150..250} costs a single GPU 18 %).  The key is the image WIDTH from the PNG header (24 bytes
  // per file, cached per sample): every rank computes the same deal without preparing the other ranks' lines; the
  // normalised length is the width scaled by the line's own height, i.e. monotone in it for lines of one source.
  vector<int> width_of(trainingset.size(), -1);
  auto sample_width = [&](int sample) {
    if (width_of[sample] < 0) {
      unsigned char h[24] = {0};
      std::ifstream f(trainingset.fnames[sample], std::ios::binary);
      f.read((char*)h, 24);
      width_of[sample] = f.gcount() == 24 ? (int)((unsigned)h[16] << 24 | (unsigned)h[17] << 16 | (unsigned)h[18] << 8 | (unsigned)h[19]) : 0;
    }
    return width_of[sample];
  };
  auto deal = [&](const vector<int>& samples) {   // positions of the minibatch this rank trains on
    vector<int> mine;
    if (ngpu <= 1) { for (int i = 0; i < batch; i++) mine.push_back(i); return mine; }
    vector<int> order(batch);
    for (int i = 0; i < batch; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return sample_width(samples[x]) > sample_width(samples[y]); });
    for (int k = ranks.rank; k < batch; k += ngpu) mine.push_back(order[k]);
    std::sort(mine.begin(), mine.end());
    return mine;
  };
  // one training sample (clstmocrtrain.cc:160-166: `lrand48() % size`, readSample)
  auto draw_one = [&](Image& raw, ustring& gt) {
    const int sample = lrand48() % trainingset.size();
    trainingset.readSample(raw, gt, sample);
  };
  // minibatch pipeline: `next` is read + normalised by a helper task while the device trains on `cur`.  The task is a
  // std::future: an exception inside it (a missing file, a character outside the codec) is re-thrown by get() on the
  // main thread, and if train_batch throws while the task is still running the future's destructor waits for it --
  // either way the process ends through main()'s "FATAL: ..." handler with exit code 1, never through std::terminate.
  // Reading a PNG and normalising the line (CenterNormalizer: Gaussian filters as wide as the line is high) costs ~30 ms
  // of host time per line -- four orders of magnitude more than the device needs for it.  So (batch > 1 only; the
  // batch = 1 loop stays the reference's, sample for sample): the minibatch's samples are DRAWN in order (same lrand48
  // sequence), then prepared on up to `prep_threads` threads, and a prepared line is kept per file (cache=1, default):
  // from the second visit of a file on, a line costs one memcpy.  Preparation is deterministic, so neither changes what
  // is trained on.
  const bool use_cache = getienv("cache", 1) != 0;
  const int prep_threads = std::max(1, std::min(getienv("prep_threads", 16), (int)std::thread::hardware_concurrency()));
  vector<std::shared_ptr<CLSTMOCR::Line>> cache(use_cache ? trainingset.size() : 0);
  auto draw = [&](CLSTMOCR::Prepared& p) {
    vector<int> samples(batch);
    for (int i = 0; i < batch; i++) samples[i] = lrand48() % trainingset.size();
    vector<std::shared_ptr<CLSTMOCR::Line>> lines(batch);
    const vector<int> mine = deal(samples);
    vector<int> todo;   // first occurrence of every file that is not cached yet
    for (int i : mine) {
      if (use_cache && cache[samples[i]]) { lines[i] = cache[samples[i]]; continue; }
      bool first = true;
      for (int j : todo) if (samples[j] == samples[i]) first = false;
      if (first) todo.push_back(i);
    }
    auto work = [&](int k0) {
      for (size_t k = k0; k < todo.size(); k += prep_threads) {
        const int i = todo[k];
        Image raw;
        ustring gt;
        trainingset.readSample(raw, gt, samples[i]);
        auto l = std::make_shared<CLSTMOCR::Line>();
        clstm.prepare_line(*l, raw, gt);
        lines[i] = l;
      }
    };
    vector<std::future<void>> pool;
    for (int t = 1; t < prep_threads && t < (int)todo.size(); t++) pool.push_back(std::async(std::launch::async, work, t));
    work(0);
    for (auto& f : pool) f.get();      // (rethrows a worker's exception)
    for (int i : todo) if (use_cache) cache[samples[i]] = lines[i];
    for (int i : mine)
      if (!lines[i]) for (int j : todo) if (samples[j] == samples[i]) lines[i] = lines[j];
    vector<const CLSTMOCR::Line*> ptrs;
    for (int i : mine) ptrs.push_back(lines[i].get());
    clstm.pack(p, ptrs);
  };
  CLSTMOCR::Prepared cur, next;
  std::future<void> helper;
  const bool batched = batch > 1 || ngpu > 1;
  if (batched) draw(next);
  for (int trial = start; trial < ntrain; trial += batch) {
    // the last trial this update covers: the triggers look at it, so that the end-of-run save / test fire for any
    // batch size (Trigger fires for good at count >= upto - 1; with batch = 8 and ntrain = 1000 the loop ends at 992)
    const int tend = std::min(trial + batch - 1, ntrain - 1);
    ustring gt, pred;
    if (!batched) {   // the reference's loop, sample for sample
      Image raw;
      draw_one(raw, gt);
      pred = clstm.train(raw, gt);
    } else {
      std::swap(cur, next);
      if (trial + batch < ntrain) helper = std::async(std::launch::async, [&] { draw(next); });
      // a minibatch whose result is printed reads decode + alignment back (synchronous); every other one is
      // enqueued without a host synchronisation and the loop goes straight on to the next
      if (report_trigger.peek(tend)) {
        vector<ustring> preds = clstm.train_batch(cur);
        pred = preds.back();
      } else {
        clstm.train_batch_async(cur);
      }
      if (helper.valid()) helper.get();
      gt = cur.targets.back();
    }
    if (report_trigger(tend)) {
      out << trial << std::endl;
      out << "TRU " << utf32_to_utf8(gt) << std::endl;
      out << "ALN " << clstm.aligned_utf8() << std::endl;
      out << "OUT " << utf32_to_utf8(pred) << std::endl;
      if (trial > 0 && report_time) out << "steptime " << (now() - start_time) / report_trigger.since() << std::endl;
      start_time = now();
    }
    if (test_trigger(tend) && testset.size() > 0 && lead) {
      double count = 0.0, errors = 0.0;
      for (int test = 0; test < testset.size(); test++) {
        Image traw;
        ustring tgt;
        testset.readSample(traw, tgt, test);
        ustring tpred = clstm.predict(traw);
        count += tgt.size();
        errors += levenshtein(tpred, tgt);
      }
      test_error = errors / count;
      std::cout << "ERROR " << trial << " " << test_error << "     " << errors << " " << count << std::endl;
      if (test_error < best_error) {
        best_error = test_error;
        string fname = save_name + ".clstm";
        std::cout << "saving best performing network so far " << fname << " error rate:  " << best_error << std::endl;
        clstm.model.attr["trial"] = std::to_string(tend);   // resume continues behind the last sample consumed
        clstm.save(fname);
      }
    }
    if (save_trigger(tend) && save_trigger.enabled) {
      string fname = save_name + "-" + std::to_string(trial) + ".clstm";
      std::cout << "saving " << fname << std::endl;
      clstm.model.attr["trial"] = std::to_string(tend);
      clstm.save(fname);
    }
  }
  return ranks.finish(0);
}

int main(int argc, char** argv) {
  try { return main1(argc, argv); }
  catch (const std::exception& e) {
    std::cerr << "FATAL: " << e.what() << std::endl;
    if (g_ranks && g_ranks->rank != 0) { fflush(nullptr); _exit(1); }   // a rank process never runs the parent's exit path
    return 1;
  }
}
