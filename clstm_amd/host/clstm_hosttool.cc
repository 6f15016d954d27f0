// clstm_hosttool -- test helper exposing the GPU-free host pieces (PNG, normaliser, model files) to
// the Python test-suite.  Raw float dumps are little-endian float32 preceded by two int32 (w, h).
#include "model.h"
#include "normalizer.h"
#include "png_io.h"
using namespace clstmhost;

static void dump(const string& fname, const Image& im) {
  std::ofstream f(fname, std::ios::binary);
  int hdr[2] = {im.w, im.h};
  f.write((const char*)hdr, 8);
  f.write((const char*)im.d.data(), im.d.size() * 4);
}

int main(int argc, char** argv) {
  try {
    string cmd = argc > 1 ? argv[1] : "";
    if (cmd == "png2raw" && argc == 4) {  // read_png
      Image im;
      read_png(im, argv[2]);
      dump(argv[3], im);
    } else if (cmd == "normalize" && argc == 5) {  // read_png, invert, CenterNormalizer -> frames
      Image im, out;
      read_png(im, argv[2]);
      for (float& v : im.d) v = -v + 1.0f;
      CenterNormalizer nz;
      nz.target_height = atoi(argv[4]);
      nz.measure(im);
      nz.normalize(out, im);
      dump(argv[3], out);
      std::cout << "r " << nz.r << " width " << out.w << std::endl;
    } else if (cmd == "writepng" && argc == 4) {  // raw -> png (write_png)
      std::ifstream f(argv[2], std::ios::binary);
      int hdr[2];
      f.read((char*)hdr, 8);
      Image im;
      im.resize(hdr[0], hdr[1]);
      f.read((char*)im.d.data(), im.d.size() * 4);
      write_png(argv[3], im);
    } else if (cmd == "init-model" && argc >= 8) {  // kind ninput nhidden nhidden2 nclasses seed out
      Model m;
      LCG lcg(atof(argv[7]));
      m.create(argv[2], atoi(argv[3]), atoi(argv[6]), atoi(argv[4]), atoi(argv[5]), lcg);
      for (int i = 0; i < m.desc.nclasses; i++) m.codec.push_back(i == 0 ? 0 : 96 + i);
      m.attr["learning_rate"] = std::to_string(1e-4);
      m.attr["momentum"] = std::to_string(0.9);
      m.save(argv[8]);
    } else if (cmd == "roundtrip" && argc == 4) {  // load model file, save it again
      Model m;
      m.load(argv[2]);
      m.save(argv[3]);
      std::cout << m.kind() << " ninput " << m.desc.ninput << " nhidden " << m.desc.nhidden[0] << " nclasses "
                << m.desc.nclasses << " nparams " << m.nparams() << std::endl;
    } else if (cmd == "params" && argc == 4) {  // flat float32 params of a model file
      Model m;
      m.load(argv[2]);
      std::ofstream f(argv[3], std::ios::binary);
      f.write((const char*)m.params.data(), m.params.size() * 4);
    } else {
      std::cerr << "usage: clstm_hosttool png2raw|normalize|writepng|init-model|roundtrip|params ...\n";
      return 2;
    }
    return 0;
  } catch (const std::exception& e) {
    std::cerr << "FATAL: " << e.what() << std::endl;
    return 1;
  }
}
