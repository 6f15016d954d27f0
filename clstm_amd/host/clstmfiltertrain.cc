// clstmfiltertrain -- the reference's text-to-text training driver (clstmfiltertrain.cc:66-158; the
// `run-cmu` configuration) on the MI355X path: tab-separated "input<TAB>output" lines, CTC-trained BiLSTM.
#include "clstmhl.h"
using namespace clstmhost;

struct Sample { ustring in, out; };

static void read_samples(vector<Sample>& samples, const string& fname) {  // :36-54
  std::ifstream stream(fname);
  if (!stream) fail("cannot open: " + fname);
  string line;
  samples.clear();
  while (getline(stream, line)) {
    if (line.substr(0, 1) == "#") continue;
    if (line.size() == 0) continue;
    size_t where = line.find("\t");
    if (where == string::npos) fail("no tab found in input line");
    ustring in = utf8_to_utf32(line.substr(0, where)), out = utf8_to_utf32(line.substr(where + 1));
    if (in.empty() || out.empty()) continue;
    samples.push_back(Sample{in, out});
  }
}
static void get_codec(vector<int>& codec, const vector<Sample>& samples, ustring Sample::*p) {  // :56-64
  std::set<int> codes;
  codes.insert(0);
  for (auto& e : samples)
    for (char32_t c : e.*p) codes.insert((int)c);
  codec.assign(codes.begin(), codes.end());
}

static int main1(int argc, char** argv) {
  if (argc < 2 || argc > 3) fail("... training [testing]");
  vector<Sample> samples, test_samples;
  read_samples(samples, argv[1]);
  if (argc > 2) read_samples(test_samples, argv[2]);
  std::cout << "got " << samples.size() << " inputs, " << test_samples.size() << " tests" << std::endl;
  if (samples.empty()) fail("no training samples");
  string load_name = getsenv("load", "");
  CLSTMText clstm;
  int nhidden = -1;
  double lrate = getdenv("lrate", 1e-4), momentum = getdenv("momentum", 0.9);
  if (load_name != "") {
    clstm.load(load_name);
  } else {
    vector<int> icodec, codec;
    get_codec(icodec, samples, &Sample::in);
    get_codec(codec, samples, &Sample::out);
    nhidden = getienv("nhidden", 100);
    clstm.createBidi(icodec, codec, nhidden);
    clstm.setLearningRate(lrate, momentum);
  }
  int ntrain = getienv("ntrain", 10000000);
  int save_every = getienv("save_every", 10000);
  string save_name = getsenv("save_name", "_filter");
  int report_every = getienv("report_every", 100);
  int test_every = getienv("test_every", 10000);
  bool use_exact = getienv("use_exact", 0);
  string after_test = getsenv("after_test", "");
  double best_error = 1e38, test_error = 9999.0;
  auto it = clstm.model.attr.find("trial");
  int start = (it == clstm.model.attr.end() ? getienv("start", -1) : atoi(it->second.c_str())) + 1;
  if (start > 0) std::cout << "start " << start << std::endl;
  for (int trial = start; trial < ntrain; trial++) {
    int sample = lrand48() % samples.size();
    if (trial > 0 && test_samples.size() > 0 && test_every > 0 && trial % test_every == 0) {
      double errors = 0.0, count = 0.0, exact = 0.0;
      for (auto& ts : test_samples) {
        ustring pred = clstm.predict(ts.in);
        count += ts.out.size();
        errors += levenshtein(pred, ts.out);
        if (pred == ts.out) exact++;
      }
      test_error = errors / count;
      double exact_test_error = 1.0 - exact / test_samples.size();
      std::cout << "ERROR " << trial << " " << test_error << "     " << errors << " " << count << " exact_errors "
                << exact_test_error << " lrate " << lrate << " momentum " << momentum << " nhidden " << nhidden << std::endl;
      if (use_exact) test_error = exact_test_error;
      if (save_every == 0 && test_error < best_error) {
        best_error = test_error;
        string fname = save_name + ".clstm";
        std::cout << "saving best performing network so far " << fname << " error rate:  " << best_error << std::endl;
        clstm.model.attr["trial"] = std::to_string(trial);
        clstm.save(fname);
      }
      if (after_test != "") (void)!system(after_test.c_str());
    }
    if (trial > 0 && save_every > 0 && trial % save_every == 0) {
      string fname = save_name + "-" + std::to_string(trial) + ".clstm";
      clstm.model.attr["trial"] = std::to_string(trial);
      clstm.save(fname);
    }
    ustring pred = clstm.train(samples[sample].in, samples[sample].out);
    if (trial % report_every == 0) {
      std::cout << "trial " << trial << std::endl;
      std::cout << "INP " << utf32_to_utf8(samples[sample].in) << std::endl;
      std::cout << "TRU " << utf32_to_utf8(samples[sample].out) << std::endl;
      std::cout << "ALN " << clstm.aligned_utf8() << std::endl;
      std::cout << "OUT " << utf32_to_utf8(pred) << std::endl;
    }
  }
  return 0;
}

int main(int argc, char** argv) {
  try { return main1(argc, argv); }
  catch (const std::exception& e) { std::cerr << "FATAL: " << e.what() << std::endl; return 1; }
}
