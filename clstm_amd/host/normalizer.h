// normalizer.h -- CenterNormalizer, the text-line normaliser in front of the hot path
// (extras.cc:53-131 gauss1d/gauss2d/bilin, :209-285 argmax1/add_smear/CenterNormalizer).
// Host preprocessing: Gaussian-smooth the line, trace its centre line as the per-column argmax,
// measure the mean absolute deviation of ink from it, and resample to `target_height` rows so that
// +-range*mad fills the height.  All arithmetic types (float/double mixes, int truncations) follow
// the cited lines so that the frames handed to set_inputs are the reference's.
#pragma once
#include "hostutil.h"

namespace clstmhost {

// acc[i] += in[i + j] * mask[j] for every tap j in the reference's order: each output keeps its own double accumulator and
// the sequence of roundings of extras.cc:76-84 (float product, double sum, taps ascending), but the outputs advance together,
// so the loop runs at vector throughput instead of one dependent double add per tap (the masks are 2*(1+3*sigma)+1 =
// 531 taps for an 88-pixel-high line: 30 ms per line otherwise).  No contraction (-ffp-contract=off in the Makefile).
__attribute__((optimize("O3"), target_clones("avx512f", "avx2", "default")))
static void gauss1d_taps(double* acc, const float* pad, const float* mask, int n, int m) {
  for (int j = 0; j < m; j++) {
    const float mj = mask[j];
    const float* p = pad + j;
    for (int i = 0; i < n; i++) acc[i] += (double)(p[i] * mj);
  }
}

inline void gauss1d(vector<float>& out, const vector<float>& in, float sigma) {  // extras.cc:58-86
  const int n = (int)in.size();
  out.assign(n, 0.0f);
  if (n == 0) return;
  const int range = 1 + int(3.0 * sigma);
  // the mask of :61-71 depends on sigma only: gauss2d asks for the same one for every column / every row of a line
  static thread_local vector<float> mask;
  static thread_local float mask_sigma = -1.0f;
  if (mask_sigma != sigma || (int)mask.size() != 2 * range + 1) {
    mask.assign(2 * range + 1, 0.0f);
    for (int i = 0; i <= range; i++) {
      double y = exp(-i * i / 2.0 / sigma / sigma);
      mask[range + i] = mask[range - i] = (float)y;
    }
    float total = 0.0f;
    for (float m : mask) total += m;
    for (float& m : mask) m /= total;
    mask_sigma = sigma;
  }
  const int m = (int)mask.size();
  // the clamped index of :79-81 as a padded copy: pad[k] = in[clamp(k - range)]
  static thread_local vector<float> pad;
  static thread_local vector<double> acc;
  pad.resize((size_t)n + 2 * range);
  for (int k = 0; k < n + 2 * range; k++) pad[k] = in[std::min(std::max(k - range, 0), n - 1)];
  acc.assign(n, 0.0);
  gauss1d_taps(acc.data(), pad.data(), mask.data(), n, m);
  for (int i = 0; i < n; i++) out[i] = (float)acc[i];
}

inline void gauss2d(Image& a, float sx, float sy) {  // extras.cc:108-121
  vector<float> r, s;
  for (int i = 0; i < a.w; i++) {      // each column (fixed x) is smoothed along y with sy
    r.assign(a.d.begin() + (size_t)i * a.h, a.d.begin() + (size_t)(i + 1) * a.h);
    gauss1d(s, r, sy);
    std::copy(s.begin(), s.end(), a.d.begin() + (size_t)i * a.h);
  }
  r.resize(a.w);
  for (int j = 0; j < a.h; j++) {      // each row (fixed y) along x with sx
    for (int i = 0; i < a.w; i++) r[i] = a(i, j);
    gauss1d(s, r, sx);
    for (int i = 0; i < a.w; i++) a(i, j) = s[i];
  }
}

inline int clipi(int x, int n) { return x < 0 ? 0 : x >= n ? n - 1 : x; }

inline float bilin(const Image& a, float x, float y) {  // extras.cc:131-143
  const int w = a.w, h = a.h;
  const int i = (int)floor(x), j = (int)floor(y);
  const float l = x - i, m = y - j;
  const float s00 = a(clipi(i, w), clipi(j, h)), s01 = a(clipi(i, w), clipi(j + 1, h));
  const float s10 = a(clipi(i + 1, w), clipi(j, h)), s11 = a(clipi(i + 1, w), clipi(j + 1, h));
  return (float)((1.0 - l) * ((1.0 - m) * s00 + m * s01) + l * ((1.0 - m) * s10 + m * s11));
}

struct CenterNormalizer {  // extras.cc:227-285
  int target_height = 48;
  float smooth2d = 1.0f, smooth1d = 0.3f, range = 4.0f;
  vector<float> center;
  float r = -1;
  void measure(const Image& line) {
    const int w = line.w, h = line.h;
    Image smooth = line;
    gauss2d(smooth, h * smooth2d, h * 0.5f);
    for (int j = 0; j < h; j++) {  // add_smear (:219-229): avoids singularities on empty columns
      double v = 0.0;
      for (int i = 0; i < w; i++) {
        v = v * 0.9 + line(i, j);
        smooth(i, j) += (float)(fmin(1.0, v) * 1e-3);
      }
    }
    vector<float> a(w);
    for (int i = 0; i < w; i++) {  // argmax1 (:205-217): ties -> last row
      float mv = smooth(i, 0), mj = 0;
      for (int j = 1; j < h; j++) {
        if (smooth(i, j) < mv) continue;
        mv = smooth(i, j);
        mj = (float)j;
      }
      a[i] = mj;
    }
    gauss1d(center, a, h * smooth1d);
    float s1 = 0.0f, sy = 0.0f;
    for (int i = 0; i < w; i++)
      for (int j = 0; j < h; j++) {
        s1 += line(i, j);
        sy += (float)(line(i, j) * fabs(j - center[i]));
      }
    const float mad = sy / s1;
    r = (float)int(range * mad + 1);
  }
  void normalize(Image& out, const Image& in) const {
    const int w = in.w;
    if (w != (int)center.size()) fail("measure doesn't match normalize");
    const float scale = (float)((2.0 * r) / target_height);
    const int target_width = std::max(int(w / scale), 1);
    out.resize(target_width, target_height);
    for (int i = 0; i < out.w; i++)
      for (int j = 0; j < out.h; j++) {
        const float x = scale * i;
        const float y = scale * (j - target_height / 2) + center[int(x)];
        out(i, j) = bilin(in, x, y);
      }
  }
};

}  // namespace clstmhost
