// hip_inetwork.h -- the fused-level INetwork adapter (VERDICT r3 "Missing 2").
//
// In the drop-in build the reference's OWN clstmhl.h, clstmocrtrain.cc and clstmocr.cc are compiled unmodified, and
// this header is what their `#include "clstm.h"` finds (integration/Makefile makes a directory of symbolic links:
// reference sources by their names, `clstm.h` -> this file, `tensor.h` -> hip_tensor.h).  It declares the part of
// clstm.h:22-330 that code is written against -- String / Assoc / Codec / INetwork / Network, make_net, set_inputs,
// sgd_update, mktargets, ctc_align_targets, trivial_decode, the model-file functions, network_info, levenshtein -- with
// the reference's names, argument meaning and error behaviour (THROW of a const char*), and NONE of its arithmetic:
// make_net("bidi" | "bidi2" | "lstm1") returns a network whose forward() / backward() are ONE clstm_net_forward /
// clstm_net_backward call each on the MI355X library (include/clstm_abi.h) for the whole sequence batch held in
// `inputs` / `outputs`, sgd_update(net) is clstm_net_update, ctc_align_targets / trivial_decode / mktargets are the
// library's CTC entry points, and the model files go through the byte-compatible codec of clstm_amd/host/proto.h.
// Definitions: hip_inetwork.cc.  Nothing here is reachable from the product library; it is the reference-side binding.
#pragma once
#include <functional>
#include <initializer_list>
#include <map>
#include <memory>
#include <string>
#include <vector>
#include "hip_tensor.h"

struct clstm_net;

namespace ocropus {
using std::function;
using std::map;
using std::pair;        // (the drivers write `pair<double, double>` unqualified: tensor.h / Eigen bring it in the reference build)
using std::shared_ptr;
using std::string;
using std::unique_ptr;
using std::vector;
using std::wstring;

typedef vector<int> Classes;          // clstm.h:30-31
typedef vector<Classes> BatchClasses;

// error path of the reference: THROW(msg) == throw (const char*) msg, caught by the drivers' CATCH(const char*)
[[noreturn]] void hip_raise(const string& msg);

// attribute values: strings that read as numbers (clstm.h:35-48)
class String : public string {
 public:
  String() {}
  String(const char* s) : string(s) {}
  String(const string& s) : string(s) {}
  String(int x) : string(std::to_string(x)) {}
  String(double x) : string(std::to_string(x)) {}
  operator double() const { return atof(c_str()); }
  double operator+() const { return atof(c_str()); }
};
// key -> value with defaults (clstm.h:51-83; no parent chain: nothing on this path sets one)
class Assoc : public map<string, String> {
 public:
  using map<string, String>::map;
  Assoc() {}
  bool contains(const string& key, bool = true) const { return find(key) != end(); }
  String get(const string& key) const {
    auto it = find(key);
    if (it == end()) hip_raise("missing parameter: " + key);
    return it->second;
  }
  String get(const string& key, String dflt) const {
    auto it = find(key);
    return it == end() ? dflt : it->second;
  }
  void set(const string& key, String value) { (*this)[key] = value; }
};

// class index <-> code point (clstm.h:86-97, clstm.cc:219-267); class 0 is the CTC blank
class Codec {
 public:
  vector<int> codec;
  map<int, int> encoder;
  int size() { return (int)codec.size(); }
  void set(const vector<int>& data);
  wchar_t decode(int cls);
  wstring decode(Classes& cs);
  void encode(Classes& cs, const wstring& s);
  void build(const vector<string>& fnames, const wstring& extra = L"");
};

class INetwork;
typedef shared_ptr<INetwork> Network;

// clstm.h:100-160.  `sub`, `states` and `parameters` exist for source compatibility and stay empty: the layers of the
// prefab live inside the device library, the parameters are ONE flat device buffer in walk_params order (n_params /
// get_params / set_params below are the reference's flat accessors, clstm.cc:859-905).
class INetwork {
 public:
  virtual ~INetwork() {}
  string kind = "";
  vector<Network> sub;
  map<string, Sequence*> states;
  map<string, Params*> parameters;
  int nseq = 0, nsteps = 0;
  virtual void setLearningRate(Float lr, Float momentum);
  Assoc attr;
  Sequence inputs, outputs;
  virtual int ninput() { return (int)(double)attr.get("ninput"); }
  virtual int noutput() { return (int)(double)attr.get("noutput"); }
  virtual void forward() = 0;
  virtual void backward() = 0;
  virtual void initialize() {}
  Codec codec, icodec;
};

Network make_net(const string& kind, const Assoc& params);      // clstm_prefab.cc:163-173
void set_inputs(Network net, Sequence& inputs);                 // clstm.cc:677-690
void set_inputs(Network net, TensorMap2 inputs);
void sgd_update(Network net);                                   // clstm.cc:201-217
int n_params(Network net);
void get_params(Network net, Float* params, int total, int gpu = -1);
void set_params(Network net, const Float* params, int total, int gpu = -1);
void get_derivs(Network net, Float* params, int total, int gpu = -1);
void network_info(Network net, string prefix);                  // clstm.cc:269-277
bool maybe_save_net(const string& file, Network net);           // clstm_proto.cc:139-180
Network maybe_load_net(const string& file);
void save_net(const string& file, Network net);
Network load_net(const string& file);

// ctc.cc:57-190 on the device
void mktargets(Sequence& seq, Classes& transcript, int ndim);
void ctc_align_targets(Sequence& posteriors, Sequence& outputs, Sequence& targets);
void ctc_align_targets(Sequence& posteriors, Sequence& outputs, Classes& targets);
void trivial_decode(Classes& cs, Sequence& outputs, int batch = 0);
void trivial_decode(Classes& cs, Sequence& outputs, int batch, vector<int>* locs);

// edit distance with unit costs (clstm.h:303-328 has it as a header template; the drivers call it on wstrings)
template <class A, class B>
double levenshtein(A& a, B& b) {
  const size_t n = a.size(), m = b.size();
  vector<double> row(m + 1);
  for (size_t j = 0; j <= m; j++) row[j] = (double)j;
  for (size_t i = 1; i <= n; i++) {
    double diag = row[0];
    row[0] = (double)i;
    for (size_t j = 1; j <= m; j++) {
      const double subst = diag + (a[i - 1] == b[j - 1] ? 0.0 : 1.0);
      diag = row[j];
      row[j] = fmin(fmin(row[j] + 1.0, row[j - 1] + 1.0), subst);
    }
  }
  return row[m];
}
}  // namespace ocropus
