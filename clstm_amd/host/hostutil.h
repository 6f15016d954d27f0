// hostutil.h -- small host utilities of the drivers: environment-variable configuration with the
// reference's "#: name = value" echo (utils.h:161-249), Trigger (utils.h:274-322), UTF-8 <-> UTF-32
// (pstring.h:12-72), file helpers (utils.h:59-103), levenshtein (clstm.h:329-351).
#pragma once
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#include <fstream>
#include <iostream>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

namespace clstmhost {
using std::string;
using std::vector;
typedef std::u32string ustring;  // the reference uses std::wstring (UTF-32 on Linux)

struct HostError : std::runtime_error {
  using std::runtime_error::runtime_error;
};
[[noreturn]] inline void fail(const string& msg) { throw HostError(msg); }

inline double now() {
  struct timeval tv;
  gettimeofday(&tv, nullptr);
  return tv.tv_sec + 1e-6 * tv.tv_usec;
}

// ---- environment configuration --------------------------------------------------------------
inline bool reported_params(const char* name) {  // clstm.cc:20-28: report each variable once
  static std::set<string> seen;
  return !seen.insert(name).second;
}
template <class T>
inline void report_params(const char* name, const T& value) {
  const char* flag = getenv("params");
  if (flag && !atoi(flag)) return;
  if (reported_params(name)) return;
  std::cerr << "#: " << name << " = " << value << std::endl;
}
inline string getsenv(const char* name, const char* dflt) {
  const char* r = getenv(name) ? getenv(name) : dflt;
  report_params(name, r);
  return r;
}
inline int getienv(const char* name, int dflt = 0) {
  int r = getenv(name) ? atoi(getenv(name)) : dflt;
  report_params(name, r);
  return r;
}
inline double getdenv(const char* name, double dflt = 0) {
  double r = getenv(name) ? atof(getenv(name)) : dflt;
  report_params(name, r);
  return r;
}
// value or log-uniform random value "lo,hi" (utils.h:214-230)
inline double getrenv(const char* name, double dflt = 0) {
  const char* s = getenv(name);
  if (!s) return dflt;
  float lo, hi;
  if (sscanf(s, "%g,%g", &lo, &hi) == 2) {
    double x = exp(log(lo) + drand48() * (log(hi) - log(lo)));
    report_params(name, x);
    return x;
  } else if (sscanf(s, "%g", &lo) == 1) {
    report_params(name, lo);
    return lo;
  }
  fail("bad format for getrenv");
}

// ---- "report every ..." logic, utils.h:274-322 ---------------------------------------------------
struct Trigger {
  bool finished = false, enabled = true;
  int count = 0, every = 1, upto = 0, next = 0, last_trigger = 0, current_trigger = 0;
  Trigger(int every_, int upto_ = -1, int start = 0) : count(start), every(every_), upto(upto_) {}
  Trigger& skip0() { next += every; return *this; }
  Trigger& enable(bool f) { enabled = f; return *this; }
  void rotate() { last_trigger = current_trigger; current_trigger = count; }
  int since() { return count - last_trigger; }
  bool check() {
    if (upto > 0 && count >= upto - 1) { finished = true; rotate(); return true; }
    if (every == 0) return false;
    if (count >= next) {
      while (count >= next) next += every;
      rotate();
      return true;
    }
    return false;
  }
  bool operator()(int current) { count = current; return check(); }
  bool peek(int current) const {   // would operator()(current) fire?  (no state change)
    return (upto > 0 && current >= upto - 1) || (every != 0 && current >= next);
  }
};

// ---- text -------------------------------------------------------------------------------------
inline ustring utf8_to_utf32(const string& s) {
  ustring out;
  size_t i = 0;
  while (i < s.size()) {
    unsigned c = (unsigned char)s[i], w;
    if ((c & 0x80) == 0) { w = c; i += 1; }
    else if ((c & 0xe0) == 0xc0) {
      if (i + 1 >= s.size()) fail("bad encoding");
      w = ((c & 0x1f) << 6) | ((unsigned char)s[i + 1] & 0x3f); i += 2;
    } else if ((c & 0xf0) == 0xe0) {
      if (i + 2 >= s.size()) fail("bad encoding");
      w = ((c & 0x0f) << 12) | (((unsigned char)s[i + 1] & 0x3f) << 6) | ((unsigned char)s[i + 2] & 0x3f); i += 3;
    } else if ((c & 0xf8) == 0xf0) {
      if (i + 3 >= s.size()) fail("bad encoding");
      w = ((c & 0x0f) << 18) | (((unsigned char)s[i + 1] & 0x3f) << 12) |
          (((unsigned char)s[i + 2] & 0x3f) << 6) | ((unsigned char)s[i + 3] & 0x3f); i += 4;
    } else fail("unicode character out of range");
    out.push_back((char32_t)w);
  }
  return out;
}
inline string utf32_to_utf8(const ustring& s) {
  string r;
  for (char32_t ch : s) {
    unsigned c = ch;
    if (c < 0x80) r.push_back(char(c));
    else if (c <= 0x7ff) { r.push_back(char((c >> 6) | 0xc0)); r.push_back(char((c & 0x3f) | 0x80)); }
    else if (c <= 0xffff) {
      r.push_back(char((c >> 12) | 0xe0)); r.push_back(char(((c >> 6) & 0x3f) | 0x80)); r.push_back(char((c & 0x3f) | 0x80));
    } else if (c <= 0x10ffff) {
      r.push_back(char((c >> 18) | 0xf0)); r.push_back(char(((c >> 12) & 0x3f) | 0x80));
      r.push_back(char(((c >> 6) & 0x3f) | 0x80)); r.push_back(char((c & 0x3f) | 0x80));
    } else fail("unicode character out of range");
  }
  return r;
}
inline string basename_noext(const string& s) {  // utils.h:59-72: strip from the first '.' of the last path part
  size_t start = 0;
  for (;;) {
    size_t pos = s.find("/", start);
    if (pos == string::npos) break;
    start = pos + 1;
  }
  size_t pos = s.find(".", start);
  return pos == string::npos ? s : s.substr(0, pos);
}
inline string read_text(const string& fname, int maxsize = 65536) {
  std::ifstream stream(fname, std::ios::binary);
  if (!stream) fail("cannot open: " + fname);
  string buf(maxsize - 1, '\0');
  stream.read(&buf[0], maxsize - 1);
  size_t n = stream.gcount();
  while (n > 0 && buf[n - 1] == '\n') n--;
  return buf.substr(0, n);
}
inline void read_lines(vector<string>& lines, const string& fname) {
  std::ifstream stream(fname);
  if (!stream) fail("cannot open: " + fname);
  string line;
  lines.clear();
  while (getline(stream, line)) lines.push_back(line);
}
inline void write_text(const string& fname, const string& data) {
  std::ofstream stream(fname);
  stream << data << std::endl;
}
template <class A>
inline double levenshtein(const A& a, const A& b) {  // clstm.h:329-351
  size_t n = a.size(), m = b.size();
  if (n > m) return levenshtein(b, a);
  vector<double> cur(n + 1), prev(n + 1);
  for (size_t k = 0; k <= n; k++) cur[k] = k;
  for (size_t i = 1; i <= m; i++) {
    prev = cur;
    std::fill(cur.begin(), cur.end(), 0.0);
    cur[0] = i;
    for (size_t j = 1; j <= n; j++) {
      double add = prev[j] + 1, del = cur[j - 1] + 1, change = prev[j - 1] + (a[j - 1] != b[i - 1] ? 1 : 0);
      cur[j] = fmin(fmin(add, del), change);
    }
  }
  return cur[n];
}

// (w x h) float image addressed image(x, y) like the reference's Tensor2 (tensor.h:252)
struct Image {
  int w = 0, h = 0;
  vector<float> d;
  void resize(int w_, int h_) { w = w_; h = h_; d.assign((size_t)w * h, 0.0f); }
  float& operator()(int x, int y) { return d[(size_t)x * h + y]; }
  float operator()(int x, int y) const { return d[(size_t)x * h + y]; }
};

}  // namespace clstmhost
