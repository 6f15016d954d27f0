// cr_math_check.cc -- CPU check of clstm_amd/csrc/cr_math.h against the host libm (glibc's expf/logf are
// correctly rounded in all but ~0.05 % of arguments).  Built and run by tests/test_cr_math.py.
//   usage: cr_math_check <n> <seed>   ->  "<n> <exp!=libm> <log!=libm> <two-call log_add != libm> <fused != libm>
//            <two-call off by >1ulp> <fused off by >1ulp> <fused != correctly rounded ln(sf)> <corner mismatches>"
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../clstm_amd/csrc/cr_math.h"

static uint64_t rng_state;
static double uniform() {  // xorshift64*
  rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
  return (double)((rng_state * 0x2545F4914F6CDD1Dull) >> 11) / 9007199254740992.0;
}
static int ulp_diff(float a, float b) {
  int32_t ia, ib;
  memcpy(&ia, &a, 4); memcpy(&ib, &b, 4);
  if (ia < 0) ia = (int32_t)0x80000000 - ia;
  if (ib < 0) ib = (int32_t)0x80000000 - ib;
  const int64_t d = (int64_t)ia - ib;
  return (int)(d < 0 ? -d : d);
}
int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 1000000;
  rng_state = argc > 2 ? strtoull(argv[2], 0, 10) : 88172645463325252ull;
  const CrTables t{CTC_EXP2_32, CTC_LOG_INVC, CTC_LOG_LOGC, &CTC_SOFTPLUS[0][0]};
  long me = 0, ml = 0, mo = 0, mn = 0, big_o = 0, big_n = 0, mcr = 0;
  for (long i = 0; i < n; i++) {
    // exp / log on their own: arguments over the whole range the CTC phases use
    const float x = (float)((uniform() * 2.0 - 1.0) * 30.0);
    if (cr_expf(x, t) != expf(x)) me++;
    const float p = (float)exp((uniform() * 2.0 - 1.0) * 12.0);
    if (cr_logf(p, t) != logf(p)) ml++;
    // log_add's transcendental part over its |d| <= 10 window (a share at the centre and at the edges)
    float d;
    const double sel = uniform();
    if (sel < 0.8) d = (float)((uniform() * 2.0 - 1.0) * 10.0);
    else if (sel < 0.9) d = (float)((uniform() * 2.0 - 1.0) * 1e-3);
    else d = (float)(10.0 - uniform() * 1e-3) * (uniform() < 0.5 ? 1.0f : -1.0f);
    const float ref = logf(expf(d) + 1.0f);                       // what the reference computes (host libm)
    const float two = cr_logf(cr_expf(d, t) + 1.0f, t);           // composition of the two CR functions
    const float one = cr_softplusf(d, t);                         // fused evaluation
    const float exact = (float)logl((long double)(cr_expf(d, t) + 1.0f));
    const int uo = ulp_diff(two, ref), un = ulp_diff(one, ref);
    if (uo) mo++;
    if (un) mn++;
    if (uo > 1) big_o++;
    if (un > 1) big_n++;
    if (one != exact) mcr++;
  }
  const float corners[] = {0.0f, -0.0f, 10.0f, -10.0f, 1e-30f, -1e-30f, 0x1p-24f, -0x1p-24f, 9.999999f, -9.999999f};
  long mc = 0;
  for (float d : corners)
    if (cr_softplusf(d, t) != (float)logl((long double)(cr_expf(d, t) + 1.0f))) mc++;
  printf("%ld %ld %ld %ld %ld %ld %ld %ld %ld\n", n, me, ml, mo, mn, big_o, big_n, mcr, mc);
  return 0;
}
