// model.h -- host-side model description: the prefab topologies of clstm_prefab.cc:23-109 ("lstm1",
// "bidi", "bidi2") as a flat parameter vector in walk_params order (clstm.cc:59-62) plus codec and
// attributes, with conversion to/from the reference's model files (clstm_proto.cc:61-137) and the
// reference's weight initialisation (clstm.cc:30-36, batches.cc:11-52).  No device code here.
#pragma once
#include <algorithm>

#include "../../include/clstm_abi.h"
#include "proto.h"

namespace clstmhost {

// batches.cc:11-17: state = frac(189843.9384938*state + 0.328340981343), seed from $seed or 0.1
struct LCG {
  double state;
  LCG() { state = getenv("seed") ? atof(getenv("seed")) : 0.1; }
  explicit LCG(double s) : state(s) {}
  double randu() {
    volatile double prod = 189843.9384938 * state;  // no fused multiply-add: keep the reference's bits
    volatile double sum = prod + 0.328340981343;
    state = sum - floor(sum);
    return state;
  }
  // column-major (n x m) block, filled i outer / j inner, mode "negbiased" (batches.cc:39-41)
  void rinit(float* a, int n, int m, float s = 0.01f, float offset = 0.0f) {
    for (int i = 0; i < n; i++)
      for (int j = 0; j < m; j++) a[i + (size_t)n * j] = (float)(3 * s * randu() - 2 * s + offset);
  }
};

struct Model {
  clstm_net_desc desc{};
  vector<float> params;  // flat, walk_params order
  vector<int> codec, icodec;
  std::map<string, string> attr;  // top-level attributes (kind, learning_rate, momentum, trial, ...)

  int ndir() const { return desc.unidirectional ? 1 : 2; }
  int nparams() const { return clstm_net_nparams_for(&desc); }
  string kind() const { return desc.unidirectional ? "lstm1" : desc.nlayers == 2 ? "bidi2" : "bidi"; }

  // make_net(kind, {ninput, noutput, nhidden[, nhidden2]}) + initialize()  (clstm_prefab.cc:163-173)
  void create(const string& kind_, int ninput, int noutput, int nhidden, int nhidden2, LCG& lcg) {
    desc = clstm_net_desc{};
    desc.ninput = ninput;
    desc.nclasses = noutput;
    desc.nhidden[0] = nhidden;
    if (kind_ == "bidi") { desc.nlayers = 1; }
    else if (kind_ == "bidi2") { desc.nlayers = 2; desc.nhidden[1] = nhidden2; }
    else if (kind_ == "lstm1") { desc.nlayers = 1; desc.unidirectional = 1; }
    else fail("no such network or layer: " + kind_ + " (MI355X path: lstm1, bidi, bidi2)");
    params.assign(nparams(), 0.0f);
    attr.clear();
    attr["kind"] = kind_;
    // construction order = initialisation order: forward NPLSTM (WGI,WGF,WGO,WCI), reversed NPLSTM,
    // next layer, softmax (clstm.cc:587-590, clstm_prefab.cc:52-68); flat order is alphabetical
    static const int flatpos[4] = {2, 1, 3, 0};  // WGI,WGF,WGO,WCI -> index among WCI,WGF,WGI,WGO
    size_t off = 0;
    int ni = ninput;
    for (int l = 0; l < desc.nlayers; l++) {
      const int no = desc.nhidden[l];
      const size_t blk = (size_t)no * (ni + no + 1);
      for (int d = 0; d < ndir(); d++) {
        for (int k = 0; k < 4; k++) lcg.rinit(&params[off + flatpos[k] * blk], no, ni + no + 1);
        off += 4 * blk;
      }
      ni = ndir() * no;
    }
    lcg.rinit(&params[off], noutput, ni + 1);
  }

  // ---- model file <-> Model -----------------------------------------------------------------------
  static ArrayProto array_of(const string& name, const float* p, int n, int m) {  // proto_of_params :35-45
    ArrayProto a;
    a.name = name;
    a.dim = {n, m};
    a.value.resize((size_t)n * m);
    for (int i = 0; i < n; i++)
      for (int j = 0; j < m; j++) a.value[(size_t)i * m + j] = p[i + (size_t)n * j];
    return a;
  }
  static void params_of(float* p, const ArrayProto& a, int n, int m) {  // params_of_proto :47-59
    if (a.dim.size() != 2) fail("bad format (Mat, " + a.name + ")");
    if (a.dim[0] != n || a.dim[1] != m) fail("weight " + a.name + " has unexpected dimensions");
    if (a.value.empty()) { std::fill(p, p + (size_t)n * m, 0.0f); return; }
    if (a.value.size() != (size_t)n * m) fail("bad size (Mat)");
    for (int i = 0; i < n; i++)
      for (int j = 0; j < m; j++) p[i + (size_t)n * j] = a.value[(size_t)i * m + j];
  }
  NetProto lstm_proto(const float* p, int ni, int no) const {
    NetProto n;
    n.kind = "NPLSTM";
    n.ninput = ni;
    n.noutput = no;
    // layer(lstm_type, ni, no, params, {}) copies the make_net arguments into the layer's attributes
    // (clstm.cc:104-108); ninput/noutput are not saved (clstm_proto.cc:81-82)
    n.attribute.emplace_back("nhidden", std::to_string(desc.nhidden[0]));
    if (desc.nlayers == 2) n.attribute.emplace_back("nhidden2", std::to_string(desc.nhidden[1]));
    const size_t blk = (size_t)no * (ni + no + 1);
    static const char* names[4] = {"WCI", "WGF", "WGI", "WGO"};
    for (int k = 0; k < 4; k++) n.weights.push_back(array_of(names[k], p + k * blk, no, ni + no + 1));
    return n;
  }
  NetProto to_proto() const {  // proto_of_net, clstm_proto.cc:61-98
    NetProto top;
    top.kind = "Stacked";
    top.ninput = desc.ninput;
    top.noutput = desc.nclasses;
    top.codec = codec;
    top.icodec = icodec;
    for (auto& kv : attr) {
      if (kv.first == "name" || kv.first == "ninput" || kv.first == "noutput") continue;
      top.attribute.emplace_back(kv.first, kv.second);
    }
    size_t off = 0;
    int ni = desc.ninput;
    for (int l = 0; l < desc.nlayers; l++) {
      const int no = desc.nhidden[l];
      const size_t blk4 = (size_t)4 * no * (ni + no + 1);
      if (desc.unidirectional) {
        top.sub.push_back(lstm_proto(&params[off], ni, no));
        off += blk4;
      } else {
        NetProto par;
        par.kind = "Parallel";
        par.ninput = ni;
        par.noutput = 2 * no;
        par.sub.push_back(lstm_proto(&params[off], ni, no));
        NetProto rev;
        rev.kind = "Reversed";
        rev.ninput = ni;
        rev.noutput = no;   // Reversed::noutput() = sub[0]->noutput() (clstm.cc:459)
        rev.sub.push_back(lstm_proto(&params[off + blk4], ni, no));
        par.sub.push_back(rev);
        top.sub.push_back(par);
        off += 2 * blk4;
      }
      ni = ndir() * no;
    }
    NetProto sm;
    sm.kind = "SoftmaxLayer";
    sm.ninput = ni;
    sm.noutput = desc.nclasses;
    sm.attribute.emplace_back("nhidden", std::to_string(desc.nhidden[0]));
    if (desc.nlayers == 2) sm.attribute.emplace_back("nhidden2", std::to_string(desc.nhidden[1]));
    sm.weights.push_back(array_of("W1", &params[off], desc.nclasses, ni + 1));
    top.sub.push_back(sm);
    return top;
  }
  static const ArrayProto& weight(const NetProto& n, const string& name) {
    for (auto& w : n.weights)
      if (w.name == name) return w;
    fail("model file: layer " + n.kind + " lacks weight " + name);
  }
  static const NetProto& the_lstm(const NetProto& n) {  // NPLSTM, or Reversed{NPLSTM}
    if (n.kind == "NPLSTM") return n;
    if (n.kind == "Reversed" && n.sub.size() == 1 && n.sub[0].kind == "NPLSTM") return n.sub[0];
    fail("model file: unsupported layer kind '" + n.kind + "' (MI355X path: NPLSTM / Reversed / Parallel / "
         "Stacked / SoftmaxLayer in the lstm1, bidi, bidi2 arrangements)");
  }
  void from_proto(const NetProto& top) {  // net_of_proto, clstm_proto.cc:100-137
    if (top.kind != "Stacked" || top.sub.size() < 2) fail("model file: expected a Stacked network");
    const NetProto& sm = top.sub.back();
    if (sm.kind != "SoftmaxLayer") fail("model file: output layer '" + sm.kind + "' is not supported (SoftmaxLayer only)");
    desc = clstm_net_desc{};
    desc.ninput = top.ninput;
    desc.nlayers = (int)top.sub.size() - 1;
    if (desc.nlayers > CLSTM_MAX_LAYERS) fail("model file: too many layers");
    desc.unidirectional = top.sub[0].kind == "NPLSTM";
    vector<const NetProto*> lstms;
    for (int l = 0; l < desc.nlayers; l++) {
      const NetProto& n = top.sub[l];
      if (desc.unidirectional) lstms.push_back(&the_lstm(n));
      else {
        if (n.kind != "Parallel" || n.sub.size() != 2 || n.sub[1].kind != "Reversed")
          fail("model file: expected Parallel{NPLSTM, Reversed{NPLSTM}}, got " + n.kind);
        lstms.push_back(&the_lstm(n.sub[0]));
        lstms.push_back(&the_lstm(n.sub[1]));
      }
    }
    // dimensions come from the weights, as GenericNPLSTM::postLoad does (clstm.cc:594-599)
    for (int l = 0; l < desc.nlayers; l++) desc.nhidden[l] = weight(*lstms[l * ndir()], "WGI").dim.at(0);
    desc.nclasses = weight(sm, "W1").dim.at(0);
    params.assign(nparams(), 0.0f);
    size_t off = 0;
    int ni = desc.ninput, li = 0;
    static const char* names[4] = {"WCI", "WGF", "WGI", "WGO"};
    for (int l = 0; l < desc.nlayers; l++) {
      const int no = desc.nhidden[l];
      const size_t blk = (size_t)no * (ni + no + 1);
      for (int d = 0; d < ndir(); d++, li++)
        for (int k = 0; k < 4; k++, off += blk) params_of(&params[off], weight(*lstms[li], names[k]), no, ni + no + 1);
      ni = ndir() * no;
    }
    params_of(&params[off], weight(sm, "W1"), desc.nclasses, ni + 1);
    codec = top.codec;
    icodec = top.icodec;
    attr.clear();
    for (auto& kv : top.attribute) attr[kv.first] = kv.second;
  }
  void save(const string& fname) const { save_proto(fname, to_proto()); }
  void load(const string& fname) { from_proto(load_proto(fname)); }
};

}  // namespace clstmhost
