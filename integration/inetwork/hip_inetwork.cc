// hip_inetwork.cc -- definitions behind hip_inetwork.h: the reference's INetwork surface over the fused C ABI of
// libclstm_hip.so (include/clstm_abi.h), plus the host functions the unmodified reference drivers link against
// (extras.h: CenterNormalizer / read_png / write_png; utils.h: reported_params).  Compiled in the directory of symbolic
// links that integration/Makefile sets up, so "extras.h" / "utils.h" / "pstring.h" below ARE the reference's files and
// "clstm.h" is hip_inetwork.h.  The arithmetic of the path is all behind the C ABI; what is host code here is the
// reference's own host side (normaliser, PNG, codec, model file) as restated in clstm_amd/host/.
#include "clstm.h"
#include "extras.h"
#include "utils.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <set>

#include "../../clstm_amd/host/clstmhl.h"   // Model (prefab layout, rinit, model file), CenterNormalizer, PNG

namespace ocropus {

[[noreturn]] void hip_raise(const string& msg) {
  static thread_local string last;   // the drivers catch `const char*` (SConstruct:43): the text must outlive the throw
  last = msg;
  THROW(last.c_str());
}
static void chk(int rc, const char* what) {
  if (rc) hip_raise(string(what) + ": " + clstm_last_error());
}
template <class F>
static auto guarded(F&& f) -> decltype(f()) {   // clstmhost:: helpers report through std::exception
  try { return f(); } catch (const std::exception& e) { hip_raise(e.what()); }
}

bool reported_params(const char* name) {   // clstm.cc:20-28
  static std::set<string> seen;
  return !seen.insert(name).second;
}

// ---- Codec (clstm.cc:219-267) --------------------------------------------------------------------------------------
void Codec::set(const vector<int>& data) {
  codec = data;
  encoder.clear();
  for (int i = 0; i < (int)codec.size(); i++) encoder.insert(std::make_pair(codec[i], i));
}
wchar_t Codec::decode(int cls) { return (wchar_t)codec.at(cls); }
wstring Codec::decode(Classes& cs) {
  wstring s;
  for (int c : cs) s.push_back((wchar_t)codec.at(c));
  return s;
}
void Codec::encode(Classes& cs, const wstring& s) {   // (the reference asserts; here: the drivers' FATAL path)
  cs.clear();
  for (wchar_t ch : s) {
    auto it = encoder.find((int)ch);
    if (it == encoder.end()) hip_raise("character not in codec: U+" + std::to_string((unsigned)ch));
    if (it->second == 0) hip_raise("transcript maps to class 0 (reserved for the CTC blank)");
    cs.push_back(it->second);
  }
}
void Codec::build(const vector<string>& fnames, const wstring& extra) {
  std::set<int> codes;
  codes.insert(0);
  for (wchar_t c : extra) codes.insert((int)c);
  for (auto& fname : fnames) {
    std::ifstream stream(fname);
    string line;
    while (getline(stream, line)) {
      if (line.substr(0, 1) == "#" || line.empty()) continue;
      for (wchar_t c : utf8_to_utf32(line)) codes.insert((int)c);
    }
  }
  set(vector<int>(codes.begin(), codes.end()));
}

void INetwork::setLearningRate(Float lr, Float momentum) {   // clstm.cc:163-166
  attr.set("learning_rate", lr);
  attr.set("momentum", momentum);
}

// ---- where a drop-in step spends its time (CLSTM_ADAPTER_TIMING=1: one line per section on stderr at exit) --------------
// The reference's drivers prepare every sample on the host (read_png + CenterNormalizer, clstmocrtrain.cc:167-172) in front
// of each forward(); this splits their `steptime` into that part and the part behind the INetwork surface.
struct AdapterClock {
  enum { FORWARD, CTC, BACKWARD, UPDATE, DECODE, NORMALIZE, PNG, NSEC };
  double ms[NSEC] = {0}; long calls[NSEC] = {0};
  bool on = false;
  AdapterClock() { const char* e = getenv("CLSTM_ADAPTER_TIMING"); on = e && atoi(e) != 0; }
  ~AdapterClock() {
    if (!on) return;
    static const char* names[NSEC] = {"forward", "ctc_align", "backward", "sgd_update", "trivial_decode", "normalizer", "read_png"};
    for (int i = 0; i < NSEC; i++)
      if (calls[i]) fprintf(stderr, "adapter_time %-14s %8ld calls %10.3f ms each %12.1f ms total\n", names[i], calls[i], ms[i] / calls[i], ms[i]);
  }
};
static AdapterClock g_clock;
struct Timed {
  int which; std::chrono::steady_clock::time_point t0;
  explicit Timed(int w) : which(w) { if (g_clock.on) t0 = std::chrono::steady_clock::now(); }
  ~Timed() {
    if (!g_clock.on) return;
    g_clock.ms[which] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    g_clock.calls[which]++;
  }
};

// ---- the network: a prefab of clstm_prefab.cc:23-109 inside the device library ----------------------------------------
class HipNetwork : public INetwork {
 public:
  clstmhost::Model model;     // geometry, flat parameters on the host when needed, model-file conversion
  clstm_net* h = nullptr;
  int T = 0, bs = 0;
  vector<float> stage;
  ~HipNetwork() override { if (h) clstm_net_destroy(h); }
  void create(const string& kind_, int ni, int no, int nh, int nh2) {
    guarded([&] {
      clstmhost::LCG lcg;   // $seed, batches.cc:11-17 -- consumed by initialize()
      model.create(kind_, ni, no, nh, nh2, lcg);
    });
    attach(false);
  }
  void attach(bool upload) {
    if (h) { clstm_net_destroy(h); h = nullptr; }
    chk(clstm_net_create(&h, &model.desc, nullptr, nullptr, nullptr), "clstm_net_create");
    if (upload) chk(clstm_net_set_params_h(h, model.params.data()), "clstm_net_set_params_h");
    push_lr();
  }
  void push_lr() {
    if (!h) return;
    const float lr = (float)(double)attr.get("learning_rate", 1e-4), mom = (float)(double)attr.get("momentum", 0.9);
    chk(clstm_net_set_learning_rate(h, lr, mom), "clstm_net_set_learning_rate");
    if (attr.contains("gradient_clip")) chk(clstm_net_set_gradient_clip(h, (float)(double)attr.get("gradient_clip")), "clstm_net_set_gradient_clip");
  }
  void setLearningRate(Float lr, Float momentum) override { INetwork::setLearningRate(lr, momentum); push_lr(); }
  // the reference draws the weights in initialize() (clstm.cc:587-590), after make_net: same LCG sequence here
  void initialize() override {
    guarded([&] {
      clstmhost::LCG lcg;
      model.create(model.kind(), model.desc.ninput, model.desc.nclasses, model.desc.nhidden[0], model.desc.nhidden[1], lcg);
    });
    chk(clstm_net_set_params_h(h, model.params.data()), "clstm_net_set_params_h");
  }
  // inputs: Sequence (ninput x bs) x T  ->  bs packed lines of T frames; ONE fused forward for all of them
  void forward() override {
    Timed timed(AdapterClock::FORWARD);
    T = inputs.size(); bs = inputs.cols();
    const int ni = inputs.rows(), nc = model.desc.nclasses;
    if (T <= 0 || bs <= 0) hip_raise("forward: empty input sequence");
    if (ni != model.desc.ninput) hip_raise("forward: input rows do not match ninput");
    stage.resize((size_t)std::max(ni, nc) * T * bs);
    for (int b = 0; b < bs; b++)
      for (int t = 0; t < T; t++)
        for (int i = 0; i < ni; i++) stage[((size_t)b * T + t) * ni + i] = inputs[t].v(i, b);
    vector<int> Ts(bs, T);
    chk(clstm_net_set_batch(h, Ts.data(), bs), "clstm_net_set_batch");
    chk(clstm_net_set_inputs_h(h, stage.data()), "clstm_net_set_inputs_h");
    chk(clstm_net_forward(h), "clstm_net_forward");
    chk(clstm_net_get_outputs_h(h, stage.data()), "clstm_net_get_outputs_h");
    outputs.resize(T, nc, bs);
    for (int b = 0; b < bs; b++)
      for (int t = 0; t < T; t++)
        for (int c = 0; c < nc; c++) outputs[t].v(c, b) = stage[((size_t)b * T + t) * nc + c];
    nseq += bs; nsteps += T * bs;
  }
  // outputs[t].d holds the deltas the caller computed (clstmhl.h:211-212): ONE fused backward
  void backward() override {
    Timed timed(AdapterClock::BACKWARD);
    const int nc = model.desc.nclasses;
    if (outputs.size() != T || outputs.cols() != bs || T <= 0) hip_raise("backward without a matching forward");
    for (int b = 0; b < bs; b++)
      for (int t = 0; t < T; t++)
        for (int c = 0; c < nc; c++) stage[((size_t)b * T + t) * nc + c] = outputs[t].d(c, b);
    chk(clstm_net_set_output_deltas_h(h, stage.data()), "clstm_net_set_output_deltas_h");
    chk(clstm_net_backward(h), "clstm_net_backward");
  }
  void update() {
    Timed timed(AdapterClock::UPDATE);
    chk(clstm_net_update(h), "clstm_net_update");
    nseq = 0; nsteps = 0;
  }
};
static HipNetwork* hip(Network& net) {
  HipNetwork* p = dynamic_cast<HipNetwork*>(net.get());
  if (!p) hip_raise("not a network of the MI355X path");
  return p;
}

Network make_net(const string& kind, const Assoc& args) {   // clstm_prefab.cc:163-173
  if (kind != "bidi" && kind != "bidi2" && kind != "lstm1")
    hip_raise("no such network or layer: " + kind + " (MI355X path: lstm1, bidi, bidi2)");
  auto net = std::make_shared<HipNetwork>();
  for (auto& kv : args) net->attr.set(kv.first, kv.second);
  net->attr.set("kind", kind);
  net->kind = "Stacked";
  net->create(kind, (int)(double)args.get("ninput"), (int)(double)args.get("noutput"), (int)(double)args.get("nhidden"),
              kind == "bidi2" ? (int)(double)args.get("nhidden2") : 0);
  return net;
}
void set_inputs(Network net, Sequence& inputs) { net->inputs.copy(inputs); }
void set_inputs(Network net, TensorMap2 image) {   // clstm.cc:684-690: image(t, i) -> inputs[t].v(i, 0)
  const int T = image.dimension(0), d = image.dimension(1);
  net->inputs.resize(T, d, 1);
  for (int t = 0; t < T; t++)
    for (int i = 0; i < d; i++) net->inputs[t].v(i, 0) = image(t, i);
}
void sgd_update(Network net) { hip(net)->update(); }
int n_params(Network net) { return clstm_net_nparams(hip(net)->h); }
void get_params(Network net, Float* params, int total, int) {
  if (total != n_params(net)) hip_raise("size mismatch in get_params");
  chk(clstm_net_get_params_h(hip(net)->h, params), "clstm_net_get_params_h");
}
void set_params(Network net, const Float* params, int total, int) {
  if (total != n_params(net)) hip_raise("size mismatch in set_params");
  chk(clstm_net_set_params_h(hip(net)->h, params), "clstm_net_set_params_h");
}
void get_derivs(Network net, Float* params, int total, int) {
  if (total != n_params(net)) hip_raise("size mismatch in get_derivs");
  chk(clstm_net_get_derivs_h(hip(net)->h, params), "clstm_net_get_derivs_h");
}
// the reference prints one line per layer of the tree (clstm.cc:269-277); the prefab's tree is known from its kind
void network_info(Network net, string prefix) {
  HipNetwork* p = hip(net);
  const Float lr = net->attr.get("learning_rate", 1e-4), mom = net->attr.get("momentum", 0.9);
  const int T = net->inputs.size(), To = net->outputs.size();
  auto line = [&](const string& path, int ni, int no) {
    std::cout << path << ": " << lr << " " << mom << " in " << T << " " << ni << " out " << To << " " << no << std::endl;
  };
  const clstm_net_desc& d = p->model.desc;
  const string top = prefix + ".Stacked";
  line(top, d.ninput, d.nclasses);
  int ni = d.ninput;
  for (int l = 0; l < d.nlayers; l++) {
    const int no = d.nhidden[l];
    if (d.unidirectional) { line(top + ".NPLSTM", ni, no); ni = no; continue; }
    line(top + ".Parallel", ni, 2 * no);
    line(top + ".Parallel.NPLSTM", ni, no);
    line(top + ".Parallel.Reversed", ni, no);
    line(top + ".Parallel.Reversed.NPLSTM", ni, no);
    ni = 2 * no;
  }
  line(top + ".SoftmaxLayer", ni, d.nclasses);
}

// ---- model files (clstm_proto.cc:61-180 through clstm_amd/host/proto.h) ------------------------------------------------
bool maybe_save_net(const string& file, Network net) {
  HipNetwork* p = hip(net);
  try {
    chk(clstm_net_get_params_h(p->h, p->model.params.data()), "clstm_net_get_params_h");
    p->model.codec = net->codec.codec;
    p->model.icodec = net->icodec.codec;
    p->model.attr.clear();
    for (auto& kv : net->attr) p->model.attr[kv.first] = kv.second;
    p->model.save(file);
    return true;
  } catch (const std::exception&) { return false; }
}
Network maybe_load_net(const string& file) {
  auto net = std::make_shared<HipNetwork>();
  try { net->model.load(file); } catch (const std::exception&) { return Network(); }
  for (auto& kv : net->model.attr) net->attr.set(kv.first, kv.second);
  net->attr.set("ninput", net->model.desc.ninput);
  net->attr.set("noutput", net->model.desc.nclasses);
  net->kind = "Stacked";
  net->codec.set(net->model.codec);
  net->icodec.set(net->model.icodec);
  net->attach(true);
  return net;
}
void save_net(const string& file, Network net) { if (!maybe_save_net(file, net)) hip_raise("could not save " + file); }
Network load_net(const string& file) {
  Network n = maybe_load_net(file);
  if (!n) hip_raise("could not load " + file);
  return n;
}

// ---- CTC: ctc.cc:57-190 through the library's entry points --------------------------------------------------------------
// The per-op CTC entry points take device pointers: pinned host memory on the GPU build (device-visible, read by the host
// code around the calls directly), plain memory on the emulator.  One buffer per role, grown on demand and kept.
static Float* ctc_alloc(size_t n) {
#ifdef CLSTM_INTEGRATION_HIP
  void* p = nullptr;
  if (hipHostMalloc(&p, n * sizeof(Float), hipHostMallocDefault) != hipSuccess) return nullptr;
  return (Float*)p;
#else
  return (Float*)malloc(n * sizeof(Float));
#endif
}
static void ctc_free(Float* p) {
#ifdef CLSTM_INTEGRATION_HIP
  if (p) (void)hipHostFree(p);
#else
  free(p);
#endif
}
struct DevArr {
  Float* p = nullptr;
  DevArr(int role, size_t n) {
    static Float* pool[4] = {nullptr, nullptr, nullptr, nullptr};
    static size_t cap[4] = {0, 0, 0, 0};
    if (n > cap[role]) {
      ctc_free(pool[role]);
      cap[role] = n + n / 2;
      pool[role] = ctc_alloc(cap[role]);
      if (!pool[role]) { cap[role] = 0; hip_raise("device allocation failed"); }
    }
    p = pool[role];
  }
};
void mktargets(Sequence& seq, Classes& transcript, int ndim) {   // ctc.cc:148-157: blank-interleaved one-hot targets
  const int L = (int)transcript.size(), S = 2 * L + 1;
  vector<int> states(S);
  clstm_mktargets(states.data(), transcript.data(), L);
  seq.resize(S, ndim, 1);
  for (int s = 0; s < S; s++) {
    if (states[s] < 0 || states[s] >= ndim) hip_raise("mktargets: class out of range");
    seq[s].v(states[s], 0) = 1.0f;
  }
}
static void align_states(Sequence& posteriors, Sequence& outputs, const vector<int>& states) {
  Timed timed(AdapterClock::CTC);
  const int T = outputs.size(), nc = outputs.rows(), S = (int)states.size();
  if (outputs.cols() != 1) hip_raise("ctc_align_targets: batch size 1 (ctc.cc:59)");
  DevArr probs(0, (size_t)T * nc), al(1, (size_t)T * nc), dz(2, (size_t)T * nc);
  for (int t = 0; t < T; t++)
    for (int c = 0; c < nc; c++) probs.p[(size_t)t * nc + c] = outputs[t].v(c, 0);
  const int line_off[2] = {0, T}, state_off[2] = {0, S};
  chk(clstm_ctc_align_batch(probs.p, dz.p, al.p, nc, line_off, states.data(), state_off, 1), "clstm_ctc_align_batch");
  chk(clstm_synchronize(), "clstm_synchronize");
  posteriors.resize(T, nc, 1);
  for (int t = 0; t < T; t++)
    for (int c = 0; c < nc; c++) posteriors[t].v(c, 0) = al.p[(size_t)t * nc + c];
}
void ctc_align_targets(Sequence& posteriors, Sequence& outputs, Sequence& targets) {
  // the targets the reference passes are mktargets' one-hot rows (clstmhl.h:208-209): read the class of each state back
  const int S = targets.size(), nc = targets.rows();
  vector<int> states(S, 0);
  for (int s = 0; s < S; s++) {
    int best = 0;
    for (int c = 1; c < nc; c++) if (targets[s].v(c, 0) > targets[s].v(best, 0)) best = c;
    states[s] = best;
  }
  align_states(posteriors, outputs, states);
}
void ctc_align_targets(Sequence& posteriors, Sequence& outputs, Classes& targets) {
  const int L = (int)targets.size();
  vector<int> states(2 * L + 1);
  clstm_mktargets(states.data(), targets.data(), L);
  align_states(posteriors, outputs, states);
}
void trivial_decode(Classes& cs, Sequence& outputs, int batch, vector<int>* locs) {   // ctc.cc:159-190
  const int T = outputs.size(), nc = outputs.rows();
  cs.clear();
  if (locs) locs->clear();
  if (T == 0) return;
  Timed timed(AdapterClock::DECODE);
  DevArr probs(3, (size_t)T * nc);
  for (int t = 0; t < T; t++)
    for (int c = 0; c < nc; c++) probs.p[(size_t)t * nc + c] = outputs[t].v(c, batch);
  const int line_off[2] = {0, T};
  vector<int> cls(T), loc(T);
  int cnt = 0;
  chk(clstm_trivial_decode_batch(probs.p, nc, line_off, 1, cls.data(), loc.data(), &cnt), "clstm_trivial_decode_batch");
  cs.assign(cls.begin(), cls.begin() + cnt);
  if (locs) locs->assign(loc.begin(), loc.begin() + cnt);
}
void trivial_decode(Classes& cs, Sequence& outputs, int batch) { trivial_decode(cs, outputs, batch, nullptr); }

// ---- extras.h: line normaliser and PNG files (extras.cc:227-285, 313-561 as restated in clstm_amd/host/) ---------------
static void to_image(clstmhost::Image& im, TensorMap2 a) {   // a(x, y), x = column of the line image
  im.resize(a.dimension(0), a.dimension(1));
  for (int x = 0; x < im.w; x++)
    for (int y = 0; y < im.h; y++) im(x, y) = a(x, y);
}
static void from_image(Tensor2& a, const clstmhost::Image& im) {
  a.resize(im.w, im.h);
  for (int x = 0; x < im.w; x++)
    for (int y = 0; y < im.h; y++) a(x, y) = im(x, y);
}
struct HostCenterNormalizer : INormalizer {
  clstmhost::CenterNormalizer nz;
  void measure(TensorMap2 line) override {
    Timed timed(AdapterClock::NORMALIZE);
    nz.target_height = target_height; nz.smooth2d = smooth2d; nz.smooth1d = smooth1d; nz.range = range;
    clstmhost::Image im;
    to_image(im, line);
    guarded([&] { nz.measure(im); });
  }
  void normalize(Tensor2& out, TensorMap2 in) override {
    Timed timed(AdapterClock::NORMALIZE);
    clstmhost::Image im, o;
    to_image(im, in);
    guarded([&] { nz.normalize(o, im); });
    from_image(out, o);
  }
};
INormalizer* make_CenterNormalizer() { return new HostCenterNormalizer(); }
INormalizer* make_Normalizer(const string& name) {
  if (name == "center") return make_CenterNormalizer();
  hip_raise("unknown normalizer name: " + name + " (MI355X path: center)");
}
void read_png(Tensor2& image, const char* name) {
  Timed timed(AdapterClock::PNG);
  clstmhost::Image im;
  guarded([&] { clstmhost::read_png(im, name); });
  from_image(image, im);
}
void write_png(const char* name, TensorMap2 image) {
  clstmhost::Image im;
  to_image(im, image);
  guarded([&] { clstmhost::write_png(name, im); });
}
}  // namespace ocropus
