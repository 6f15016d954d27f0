// clstmfilter -- the reference's text-to-text prediction driver (clstmfilter.cc:32-62).
#include "clstmhl.h"
using namespace clstmhost;

static int main1(int argc, char** argv) {
  if (argc != 2) fail("give text file as an argument");
  string load_name = getsenv("load", "");
  if (load_name == "") fail("must give load= parameter");
  CLSTMText clstm;
  clstm.load(load_name);
  string line;
  std::ifstream stream(argv[1]);
  int output = getienv("output", 0);
  while (getline(stream, line)) {
    string orig = line;
    size_t where = line.find("\t");
    if (where != string::npos) line = line.substr(0, where);
    string out = clstm.predict_utf8(line);
    if (output == 0) std::cout << out << std::endl;
    else if (output == 1) std::cout << line << "\t" << out << std::endl;
    else if (output == 2) std::cout << orig << "\t" << out << std::endl;
  }
  return 0;
}

int main(int argc, char** argv) {
  try { return main1(argc, argv); }
  catch (const std::exception& e) { std::cerr << "FATAL: " << e.what() << std::endl; return 1; }
}
