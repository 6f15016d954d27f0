// clstmocr -- the reference's recognition driver (clstmocr.cc:40-117) on the MI355X path.
#include "clstmhl.h"
using namespace clstmhost;

static float scaled_log(float x) {  // clstmocr.cc:33-40 (float arithmetic, clamped to [0, 1])
  const float thresh = 10.0f;
  if (x <= 0.0f) return 0.0f;
  float l = logf(x);
  if (l < -thresh) return 0.0f;
  if (l > 0) return 1.0f;
  return (l + thresh) / thresh;
}

static int main1(int argc, char** argv) {
  if (argc != 2 || !strcmp(argv[1], "-h") || !strcmp(argv[1], "--help")) {
    std::cerr << "Usage: [VAR=VAL...] " << argv[0] << " IMAGEFILE-LIST\n  Variables: load (required) conf output save_text\n";
    return EXIT_FAILURE;
  }
  string load_name = getsenv("load", "");
  if (load_name == "") fail("must give load= parameter");
  CLSTMOCR clstm;
  clstm.load(load_name);
  bool conf = getienv("conf", 0);
  string output = getsenv("output", "text");
  bool save_text = getienv("save_text", 1);
  std::ifstream stream(argv[1]);
  string line;
  while (getline(stream, line)) {
    Image raw;
    string basename = line.substr(0, line.find_last_of("."));
    read_png(raw, line);
    for (float& v : raw.d) v = -v + 1.0f;
    if (!conf) {
      string out = clstm.predict_utf8(raw);
      std::cout << line << "\t" << out << std::endl;
      if (save_text) write_text(basename + ".txt", out);
    } else {
      std::cout << "file " << line << std::endl;
      vector<CharPrediction> preds;
      clstm.predict(preds, raw);
      for (auto& p : preds) {
        ustring c(1, p.c);
        std::cout << p.i << "\t" << p.x << "\t" << utf32_to_utf8(c) << "\t" << p.p << std::endl;
      }
    }
    if (output == "text") {
    } else if (output == "logs" || output == "posteriors") {
      Image outputs;
      clstm.get_outputs(outputs);
      if (output == "logs")
        for (float& v : outputs.d) v = scaled_log(v);
      write_png(basename + (output == "logs" ? ".lp.png" : ".p.png"), outputs);
    } else fail("unknown output format");
  }
  return 0;
}

int main(int argc, char** argv) {
  try { return main1(argc, argv); }
  catch (const std::exception& e) { std::cerr << "FATAL: " << e.what() << std::endl; return 1; }
}
