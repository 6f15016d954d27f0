// dbgopt.h -- the library's EXPERIMENT switches: kept paths against measured losers, A/B builds of one process.
// None is a product setting (DESIGN.md 9 lists those: a dozen environment variables).  An option is set by the tests through
// clstm_debug_set_option(name, value) -- they compare two kernels bit for bit within one process -- or, for A/B runs of a
// whole program, through ONE environment variable read when the library is first used:
//     CLSTM_DEBUG="gemm_stag=1,bwd_c32=0"
// Options (default): gemm_stag (2: operand tiles of the bf16-source GEMMs by LDS-DMA + staggered wave groups; 1: register-staged,
// staggered; 0: the one-barrier loop of round 4), bwd_c32 (1; 0: the 16-cell persistent backward kernel), rec_x3 (1; 0: the exact-
// f32 persistent backward recurrence on the f32 MFMA), pack_tiles (1; 0: the five single-purpose repack kernels), fuse_wx (1: the
// input projection of layers of <= 128 inputs inside the persistent forward kernel; 0: never; 2: every eligible layer), gemm_b16mc
// (1; 0: weight gradient of wide layers from f32 source rows), wide_graph (1; 0: per-step launch loops not captured into a
// hipGraph), dw_x3 / gemm_x3 (1; 0: the f32 MFMA for the fused launch's weight-gradient items / the softmax layer's backward
// pair -- what clstm_net_set_strict_f32 selects per net), update_repack (1; 0: the fused update of a one-call training step leaves
// the packed parameter copies of a narrow layer to the next step's ingest launch).
// Round 6: fwd_mfma / bwd_mfma (0 never, 1 from 640 lines, 2 always: the batched-MFMA narrow recurrences), bwd_mfma_rows,
// bwd_mfma_fused, split_terms (3; 2: two-term split of the backward products), ctc_float (0; 1: float-only log_add),
// dw_slab_tiles / dw_chunk / dw_tail_parts / dw_tail_chunks (weight-gradient slab geometry), dw_ilv (0; 1: conversion between the
// MFMAs of the weight-gradient items -- measured slower), x3_coal / x3_ilv (1 / 1: gemm_x3_128_kernel's row-per-load mapping and
// interleaved conversion; 0: the round-4 forms).
#pragma once
#include <cstdlib>
#include <map>
#include <string>

namespace clstm {
inline std::map<std::string, int>& dbg_opts() {
  static std::map<std::string, int> m = [] {
    std::map<std::string, int> r;
    const char* e = getenv("CLSTM_DEBUG");
    if (e) {
      std::string s(e);
      size_t i = 0;
      while (i < s.size()) {
        size_t j = s.find(',', i);
        if (j == std::string::npos) j = s.size();
        const std::string kv = s.substr(i, j - i);
        const size_t eq = kv.find('=');
        if (eq != std::string::npos && eq > 0) r[kv.substr(0, eq)] = atoi(kv.c_str() + eq + 1);
        i = j + 1;
      }
    }
    return r;
  }();
  return m;
}
inline int dbg_opt(const char* name, int dflt) {
  const auto& m = dbg_opts();
  const auto it = m.find(name);
  return it == m.end() ? dflt : it->second;
}
}  // namespace clstm
