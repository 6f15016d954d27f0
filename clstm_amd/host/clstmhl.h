// clstmhl.h -- CLSTMOCR (clstmhl.h:146-272) and Codec (clstm.h:82-93, clstm.cc:219-263) on top of the
// C ABI of libclstm_hip.so.  Same members and call order as the reference: measure/normalize ->
// set_inputs -> forward -> mktargets/ctc_align_targets -> backward -> trivial_decode -> sgd_update.
// The host holds only small arrays (the line image, transcripts, decodes); all network arithmetic
// runs on the MI355X.
#pragma once
#include "model.h"
#include <future>
#include <thread>
#include "normalizer.h"
#include "png_io.h"

namespace clstmhost {

typedef vector<int> Classes;

inline void chk(int rc, const char* what) {
  if (rc) fail(string(what) + ": " + clstm_last_error());
}

struct Codec {
  vector<int> codec;
  std::map<int, int> encoder;
  int size() const { return (int)codec.size(); }
  void set(const vector<int>& a) {
    codec = a;
    encoder.clear();
    for (int i = 0; i < (int)codec.size(); i++) encoder.insert(std::make_pair(codec[i], i));
  }
  void encode(Classes& cs, const ustring& s) const {  // clstm.cc:226-236 (asserts become errors)
    cs.clear();
    for (char32_t ch : s) {
      auto it = encoder.find((int)ch);
      if (it == encoder.end()) fail("character not in codec: U+" + std::to_string((unsigned)ch));
      if (it->second == 0) fail("transcript maps to class 0 (reserved for blank)");
      cs.push_back(it->second);
    }
  }
  ustring decode(const Classes& cs) const {
    ustring s;
    for (int c : cs) s.push_back((char32_t)codec.at(c));
    return s;
  }
  void build(const vector<string>& fnames, const ustring& extra = U"") {  // clstm.cc:246-267
    std::set<int> codes;
    codes.insert(0);
    for (char32_t c : extra) codes.insert((int)c);
    for (auto& fname : fnames) {
      std::ifstream stream(fname);
      string line;
      while (getline(stream, line)) {
        if (line.substr(0, 1) == "#") continue;
        if (line.size() == 0) continue;
        for (char32_t c : utf8_to_utf32(line)) codes.insert((int)c);
      }
    }
    set(vector<int>(codes.begin(), codes.end()));
  }
};

// trivial_decode (ctc.cc:159-190) on a host matrix [T][nc]; used for the "ALN" report line only
inline void trivial_decode_host(Classes& cs, const float* out, int T, int nc) {
  cs.clear();
  float mv = 0;
  int mc = -1;
  for (int t = 0; t < T; t++) {
    const float* p = out + (size_t)t * nc;
    int index = -1;
    float best = p[0];
    for (int i = 0; i < nc; i++) {
      if (p[i] < best) continue;
      index = i;
      best = p[i];
    }
    if (index == 0) {
      if (mc != -1 && mc != 0) cs.push_back(mc);
      mv = 0;
      mc = -1;
      continue;
    }
    if (best > mv) { mv = best; mc = index; }
  }
}

struct CharPrediction { int i, x; char32_t c; float p; };

// predict(): the forward pass belongs to no training step -- a non-finite logit there must not arm the device flag that blocks
// updates (clstm_net_set_training; the reference only asserts in backward, clstm.cc:630-649)
struct InferenceScope {
  clstm_net* net;
  explicit InferenceScope(clstm_net* n) : net(n) { clstm_net_set_training(net, 0); }
  ~InferenceScope() { clstm_net_set_training(net, 1); }
};

struct CLSTMOCR {
  Model model;
  Codec codec;
  CenterNormalizer normalizer;
  clstm_net* net = nullptr;
  int target_height = 48;
  int nclasses = -1;
  Image image;
  vector<float> aligned;  // [T][nclasses] of the last fwdbwd
  int T = 0;

  ~CLSTMOCR() { if (net) clstm_net_destroy(net); }
  void attach() {  // device network from the host model
    if (net) { clstm_net_destroy(net); net = nullptr; }
    chk(clstm_net_create(&net, &model.desc, nullptr, nullptr, nullptr), "clstm_net_create");
    chk(clstm_net_set_params_h(net, model.params.data()), "clstm_net_set_params_h");
    nclasses = model.desc.nclasses;
    target_height = model.desc.ninput;
    normalizer.target_height = target_height;
    codec.set(model.codec);
    const float lr = atof(attr_get("learning_rate", "1e-4").c_str());
    const float mom = atof(attr_get("momentum", "0.9").c_str());
    chk(clstm_net_set_learning_rate(net, lr, mom), "clstm_net_set_learning_rate");
  }
  string attr_get(const string& k, const string& dflt) const {
    auto it = model.attr.find(k);
    return it == model.attr.end() ? dflt : it->second;
  }
  void setLearningRate(float lr, float mom) {  // INetwork::setLearningRate, clstm.cc:158-161
    model.attr["learning_rate"] = std::to_string((double)lr);   // String(double) = std::to_string
    model.attr["momentum"] = std::to_string((double)mom);
    if (net) chk(clstm_net_set_learning_rate(net, lr, mom), "clstm_net_set_learning_rate");
  }
  void createBidi(const vector<int>& codec_, int nhidden) {  // clstmhl.h:191-200
    LCG lcg;
    model.create("bidi", target_height, (int)codec_.size(), nhidden, 0, lcg);
    model.codec = codec_;
    attach();
  }
  void createBidi2(const vector<int>& codec_, int nhidden, int nhidden2) {  // make_net("bidi2", ...), clstm_prefab.cc:86-109
    LCG lcg;
    model.create("bidi2", target_height, (int)codec_.size(), nhidden, nhidden2, lcg);
    model.codec = codec_;
    attach();
  }
  void load(const string& fname) {  // clstmhl.h:157-175
    model.load(fname);
    attach();
  }
  void save(const string& fname) {
    chk(clstm_net_get_params_h(net, model.params.data()), "clstm_net_get_params_h");
    model.save(fname);
  }
  void set_line(const Image& raw) {  // measure / normalize / set_inputs, clstmhl.h:202-205
    normalizer.measure(raw);
    normalizer.normalize(image, raw);
    T = image.w;
    chk(clstm_net_set_batch(net, &T, 1), "clstm_net_set_batch");
    chk(clstm_net_set_inputs_h(net, image.d.data()), "clstm_net_set_inputs_h");  // image(t,i) is frame-major
  }
  Classes decode_outputs(vector<int>* where = nullptr) {
    vector<int> cls(T), loc(T);
    int cnt = 0;
    chk(clstm_net_decode(net, cls.data(), loc.data(), &cnt), "clstm_net_decode");
    cls.resize(cnt);
    if (where) where->assign(loc.begin(), loc.begin() + cnt);
    return cls;
  }
  ustring fwdbwd(const Image& raw, const ustring& target) {  // clstmhl.h:201-217
    set_line(raw);
    chk(clstm_net_forward(net), "clstm_net_forward");
    Classes transcript;
    codec.encode(transcript, target);
    const int L = (int)transcript.size();
    aligned.resize((size_t)T * nclasses);
    chk(clstm_net_ctc(net, transcript.data(), &L, aligned.data()), "clstm_net_ctc");
    chk(clstm_net_backward(net), "clstm_net_backward");
    return codec.decode(decode_outputs());
  }
  void update() { chk(clstm_net_update(net), "clstm_net_update"); }
  ustring train(const Image& raw, const ustring& target) {
    ustring r = fwdbwd(raw, target);
    update();
    return r;
  }
  // ---- a minibatch of lines per update (the batched MI355X path: all lines through ONE forward / CTC / backward,
  // Params.d receives the SUM of the lines' gradients exactly as consecutive CLSTMOCR::fwdbwd calls without an
  // update in between would leave it, clstmhl.h:201-217) ----------------------------------------------------
  struct Prepared {          // host-side half of a minibatch: normalised frames + encoded transcripts
    vector<int> T, L;
    vector<float> frames;    // packed line after line, frame-major
    Classes labels;          // packed transcripts
    vector<ustring> targets;
  };
  // GPU-free: may run on a helper thread while the device works on the previous minibatch (uses only its own
  // normaliser; `codec` is read-only after attach())
  struct Line {              // one normalised line + its encoded transcript (what a dataset cache keeps per file)
    Image frames;            // (w = T, h = target_height)
    Classes labels;
    ustring target;
  };
  void prepare_line(Line& out, const Image& raw, const ustring& target) const {   // (thread-safe: own normaliser)
    CenterNormalizer nz;
    nz.target_height = target_height;
    nz.measure(raw);
    nz.normalize(out.frames, raw);
    codec.encode(out.labels, target);
    out.target = target;
  }
  void pack(Prepared& p, const vector<const Line*>& lines) const {
    p.T.clear(); p.L.clear(); p.frames.clear(); p.labels.clear(); p.targets.clear();
    for (const Line* l : lines) {
      p.T.push_back(l->frames.w);
      p.frames.insert(p.frames.end(), l->frames.d.begin(), l->frames.d.end());
      p.L.push_back((int)l->labels.size());
      p.labels.insert(p.labels.end(), l->labels.begin(), l->labels.end());
      p.targets.push_back(l->target);
    }
  }
  void prepare(Prepared& p, const vector<Image>& raws, const vector<ustring>& targets) const {
    vector<Line> lines(raws.size());
    vector<const Line*> ptrs;
    for (size_t b = 0; b < raws.size(); b++) { prepare_line(lines[b], raws[b], targets[b]); ptrs.push_back(&lines[b]); }
    pack(p, ptrs);
  }
  vector<ustring> train_batch(const Prepared& p) {
    const int bs = (int)p.T.size();
    chk(clstm_net_set_batch(net, p.T.data(), bs), "clstm_net_set_batch");
    chk(clstm_net_set_inputs_h(net, p.frames.data()), "clstm_net_set_inputs_h");
    chk(clstm_net_forward(net), "clstm_net_forward");
    int N = 0;
    for (int t : p.T) N += t;
    vector<int> cls(N), loc(N), cnt(bs);
    chk(clstm_net_decode(net, cls.data(), loc.data(), cnt.data()), "clstm_net_decode");
    aligned.resize((size_t)N * nclasses);
    Classes dummy(1, 1);
    chk(clstm_net_ctc(net, p.labels.empty() ? dummy.data() : p.labels.data(), p.L.data(), aligned.data()), "clstm_net_ctc");
    chk(clstm_net_backward(net), "clstm_net_backward");
    chk(clstm_net_update(net), "clstm_net_update");
    vector<ustring> out;
    int o = 0;
    for (int b = 0; b < bs; b++) {
      out.push_back(codec.decode(Classes(cls.begin() + o, cls.begin() + o + cnt[b])));
      o += p.T[b];
    }
    T = p.T.back();   // aligned_utf8() reports the last line of the minibatch
    if (bs > 1) aligned.erase(aligned.begin(), aligned.begin() + (size_t)(N - T) * nclasses);
    return out;
  }
  // The same minibatch without any host synchronisation (clstm_net_train_step_h: frames on a copy stream, every kernel
  // and the update enqueued): the training loop calls this for the minibatches whose result nobody prints, and
  // train_batch() -- which reads the decode and the alignment back -- where a report is due.  `p` must stay untouched
  // until the call after the next one (the frames are pageable memory: the library copies them before it returns).
  void train_batch_async(const Prepared& p) {
    Classes dummy(1, 1);
    chk(clstm_net_train_step_h(net, p.T.data(), (int)p.T.size(), p.frames.data(), p.labels.empty() ? dummy.data() : p.labels.data(),
                               p.L.data()), "clstm_net_train_step_h");
  }
  void synchronize() { chk(clstm_synchronize(), "clstm_synchronize"); }
  string aligned_utf8() {  // clstmhl.h:224-229
    Classes cs;
    trivial_decode_host(cs, aligned.data(), T, nclasses);
    return utf32_to_utf8(codec.decode(cs));
  }
  ustring predict(const Image& raw, vector<int>* where = nullptr) {  // clstmhl.h:233-242
    set_line(raw);
    InferenceScope inference(net);
    chk(clstm_net_forward(net), "clstm_net_forward");
    return codec.decode(decode_outputs(where));
  }
  string predict_utf8(const Image& raw) { return utf32_to_utf8(predict(raw)); }
  void get_outputs(Image& out) {  // [T][nclasses]
    out.resize(T, nclasses);
    chk(clstm_net_get_outputs_h(net, out.d.data()), "clstm_net_get_outputs_h");
  }
  void predict(vector<CharPrediction>& preds, const Image& raw) {  // clstmhl.h:243-262
    vector<int> where;
    set_line(raw);
    InferenceScope inference(net);
    chk(clstm_net_forward(net), "clstm_net_forward");
    Classes cs = decode_outputs(&where);
    Image out;
    get_outputs(out);
    preds.clear();
    for (int i = 0; i < (int)cs.size(); i++)
      preds.push_back(CharPrediction{i, where[i], (char32_t)codec.codec[cs[i]], out(where[i], cs[i])});
  }
};

// CLSTMText (clstmhl.h:24-144): text in, text out through the same BiLSTM + CTC path; characters are
// one-hot frames separated by `neps` all-zero frames (setInputs, clstmhl.h:83-100).
struct CLSTMText {
  Model model;
  Codec codec, icodec;
  clstm_net* net = nullptr;
  int nclasses = -1, iclasses = -1;
  int neps = 3;  // the reference never overrides it: maybe_load() shadows the member (clstmhl.h:53)
  vector<float> frames, aligned;
  int T = 0;
  ~CLSTMText() { if (net) clstm_net_destroy(net); }
  void attach() {
    if (net) { clstm_net_destroy(net); net = nullptr; }
    chk(clstm_net_create(&net, &model.desc, nullptr, nullptr, nullptr), "clstm_net_create");
    chk(clstm_net_set_params_h(net, model.params.data()), "clstm_net_set_params_h");
    nclasses = model.desc.nclasses;
    iclasses = model.desc.ninput;
    codec.set(model.codec);
    icodec.set(model.icodec);
    auto get = [&](const char* k, const char* d) { auto it = model.attr.find(k); return it == model.attr.end() ? string(d) : it->second; };
    chk(clstm_net_set_learning_rate(net, atof(get("learning_rate", "1e-4").c_str()), atof(get("momentum", "0.9").c_str())),
        "clstm_net_set_learning_rate");
  }
  void setLearningRate(float lr, float mom) {
    model.attr["learning_rate"] = std::to_string((double)lr);
    model.attr["momentum"] = std::to_string((double)mom);
    if (net) chk(clstm_net_set_learning_rate(net, lr, mom), "clstm_net_set_learning_rate");
  }
  void createBidi(const vector<int>& icodec_, const vector<int>& codec_, int nhidden) {  // clstmhl.h:70-82
    LCG lcg;
    model.create("bidi", (int)icodec_.size(), (int)codec_.size(), nhidden, 0, lcg);
    model.attr["neps"] = std::to_string(neps);
    model.icodec = icodec_;
    model.codec = codec_;
    attach();
  }
  void load(const string& fname) { model.load(fname); attach(); }
  void save(const string& fname) {
    chk(clstm_net_get_params_h(net, model.params.data()), "clstm_net_get_params_h");
    model.save(fname);
  }
  void setInputs(const ustring& s) {
    Classes cs;
    icodec.encode(cs, s);
    T = (int)cs.size() * (neps + 1) + neps;
    frames.assign((size_t)T * iclasses, 0.0f);
    int index = neps;
    for (int c : cs) {
      frames[(size_t)index * iclasses + c] = 1.0f;
      index += neps + 1;
    }
    chk(clstm_net_set_batch(net, &T, 1), "clstm_net_set_batch");
    chk(clstm_net_set_inputs_h(net, frames.data()), "clstm_net_set_inputs_h");
  }
  ustring decode_outputs() {
    vector<int> cls(T), loc(T);
    int cnt = 0;
    chk(clstm_net_decode(net, cls.data(), loc.data(), &cnt), "clstm_net_decode");
    cls.resize(cnt);
    return codec.decode(cls);
  }
  ustring train(const ustring& in, const ustring& target) {  // clstmhl.h:103-117
    setInputs(in);
    chk(clstm_net_forward(net), "clstm_net_forward");
    Classes transcript;
    codec.encode(transcript, target);
    const int L = (int)transcript.size();
    aligned.resize((size_t)T * nclasses);
    chk(clstm_net_ctc(net, transcript.data(), &L, aligned.data()), "clstm_net_ctc");
    chk(clstm_net_backward(net), "clstm_net_backward");
    ustring out = decode_outputs();   // decode before the update, as the reference's outputs are
    chk(clstm_net_update(net), "clstm_net_update");
    return out;
  }
  ustring predict(const ustring& in) {
    setInputs(in);
    InferenceScope inference(net);
    chk(clstm_net_forward(net), "clstm_net_forward");
    return decode_outputs();
  }
  string predict_utf8(const string& in) { return utf32_to_utf8(predict(utf8_to_utf32(in))); }
  string aligned_utf8() {
    Classes cs;
    trivial_decode_host(cs, aligned.data(), T, nclasses);
    return utf32_to_utf8(codec.decode(cs));
  }
};

}  // namespace clstmhost
