// lstm_bwd_dw.h -- the backward recurrence and its weight-gradient GEMM as ONE launch with two workgroup roles.
//
// The recurrence occupies 2 x lines workgroups (128 at the bench minibatch) of a 256-CU chip for ~95 us while the
// split-K weight-gradient GEMM behind it (gemm_dw.h: its slabs follow the recurrence's iterations) needs the whole
// chip for ~52 us.  Running the GEMM on a second stream costs more than it saves (fork and join through events
// ~37 us, profiles/r02_timeline_overlap.txt).  Here both live in one grid: blocks [0, nrec) walk their line exactly as
// lstm_bwd_kernel does -- they are dispatched first, never wait for anything and run at raised priority -- the block
// behind them is the monitor, and every further block computes ONE (slab, output tile) item of the GEMM as soon as every
// line reports the slab's iterations complete (gemm_dw_body).  (Persistent workers pulling items from per-XCD queues were
// measured slower -- 0.362 vs 0.356 ms per step: a 448-thread slot per worker leaves room for two on an idle CU.)
// PRODUCERS (top layer): the deltas on this layer's outputs, dH = the softmax layer's x.d = W1[:,1:]^T z.d
// (backward_softmax, clstm_compute.cc:346-356), used to be a GEMM launch of its own between the CTC kernel and this one
// (12-13 us).  Now the workgroups right behind the recurrence's compute it chunk by chunk in the order the recurrence
// consumes it (wave_tiles.h: 32 frames x no columns per item, K = classes, exact f32 MFMA), the recurrence workgroups
// compute their own first 16 iterations' rows themselves, and wave 3 of each checks the chunk flags two steps ahead of
// its loads -- the scheme of the forward launch (lstm_fwd_fused.h).
#pragma once
#include "gemm_dw.h"
#include "lstm_seq.h"
#include "wave_tiles.h"

namespace clstm {

constexpr int BWD_FT = 2;   // 16-frame tiles per producer item
struct BwdProdArgs {
  const float* Dz; int nc; long long dz_elems;   // output deltas of the softmax layer [N][nc]
  const float* W1; long long w1_elems;           // softmax Params, column-major nc x (1 + ndir * no): W1[c + nc * (1 + k)]
  float* dH;                                     // [N][ndir * no]
  const int* pitems; int npitems;                // time order from chunk 1, 4 ints each: (line << 13 | dir << 12 | chunk), line offset, T, 0
  int* hflag;                                    // = LstmSeqArgs::gflag: [ndir][bs][gchunks]
  int npb;                                       // producer workgroups (= npitems)
};
DEVFN WaveTileProblem bwd_dh_problem(const LstmSeqArgs& a, const BwdProdArgs& h, const int dir) {
  WaveTileProblem p;
  p.xbuf = make_buf(h.Dz, (size_t)h.dz_elems * 4); p.ldx = h.nc;
  // rows k = dir * no .. of the transposed weight matrix: W1[:, 1 + k] is contiguous over the classes
  p.wbuf = make_buf(h.W1 + (size_t)h.nc * (1 + dir * a.no), (size_t)a.no * h.nc * 4); p.ldw = h.nc;
  p.ng = (h.nc + 15) >> 4; p.kvalid = h.nc;
  p.bbuf = make_buf(h.W1, 0);
  p.obuf = make_buf(h.dH, (size_t)a.line_off[a.bs] * a.ndir * a.no * 4); p.ldo = a.ndir * a.no; p.ocol0 = dir * a.no;
  p.ncols = a.no; p.ntiles = (a.no + 15) >> 4;
  return p;
}
// iterations [16 c, 16 c + 16) of the BACKWARD pass: direction 0 visits frame T - 1 - it, direction 1 frame it
DEVFN void bwd_self_produce(const LstmSeqArgs& a, const BwdProdArgs& h, const int dir, const int off, const int T) {
  const int wave = wave_uniform((int)threadIdx.x >> 6), nw = (int)blockDim.x >> 6;
  const int f0 = dir == 0 ? T - 16 : 0;
  const long long fbase[1] = {(long long)off + f0};
  const int flo[1] = {f0 < 0 ? -f0 : 0}, fhi[1] = {T - f0 < 16 ? T - f0 : 16};
  wave_tiles<1>(bwd_dh_problem(a, h, dir), fbase, flo, fhi, wave, nw);
}
DEVFN void bwd_dh_item(const LstmSeqArgs& a, const BwdProdArgs& h, const int it) {
  const int wave = wave_uniform((int)threadIdx.x >> 6), nw = (int)blockDim.x >> 6;
  struct alignas(16) ItemRec { int code, off, T, pad; };
  const ItemRec rec = reinterpret_cast<const ItemRec*>(h.pitems)[it];
  const int b = rec.code >> 13, dir = (rec.code >> 12) & 1, c0 = rec.code & 4095, off = rec.off, T = rec.T;
  const int nchunk = (T + 15) >> 4;
  long long fbase[BWD_FT];
  int flo[BWD_FT], fhi[BWD_FT];
#pragma unroll
  for (int t = 0; t < BWD_FT; t++) {
    const int f0 = dir == 0 ? T - 16 * (c0 + t) - 16 : 16 * (c0 + t);
    fbase[t] = (long long)off + f0;
    flo[t] = f0 < 0 ? -f0 : 0;
    fhi[t] = c0 + t < nchunk ? (T - f0 < 16 ? T - f0 : 16) : 0;
  }
  wave_tiles<BWD_FT>(bwd_dh_problem(a, h, dir), fbase, flo, fhi, wave, nw);
  drain_vmem();      // every storing wave: its rows are in memory ...
  __syncthreads();
  if (threadIdx.x == 0) {   // ... before the flags are
#pragma unroll
    for (int t = 0; t < BWD_FT; t++)
      if (c0 + t < nchunk) store_i32_wt(h.hflag + ((size_t)dir * a.bs + b) * a.gchunks + c0 + t, a.gepoch);
  }
}

struct BwdDwKernelArgs { LstmSeqArgs a; GemmDwArgs g; BwdProdArgs h; int nrec; };
template <int NK4, int KU, bool PROD>
__global__ __launch_bounds__(64 * NK4) CLSTM_TWO_WAVES_PER_SIMD void lstm_bwd_dw_kernel(BwdDwKernelArgs k) {
  const LstmSeqArgs& a = k.a;
  const GemmDwArgs& g = k.g;
  const int nrec = k.nrec;
  __shared__ __attribute__((aligned(16))) float gsm[DW_SMEM_FLOATS];
  if ((int)blockIdx.x < nrec) {
#ifndef CLSTM_HIP_EMU
    __builtin_amdgcn_s_setprio(3);
#endif
    const long long t0 = g.trace ? wall_clock() : 0;
    const int bl = (int)blockIdx.x % a.bs;
    lstm_bwd_body<NK4, KU, PROD>(a, a.order ? a.order[bl] : bl, (int)blockIdx.x / a.bs, &k.h);
    if (g.done && threadIdx.x == 0) atomic_add_i32(g.done, 1);   // (the body ended with drain + barrier: this line's deltas are in memory)
    if (g.trace && threadIdx.x == 0) { g.trace[blockIdx.x * 4] = t0; g.trace[blockIdx.x * 4 + 2] = wall_clock(); }
  } else if (PROD && (int)blockIdx.x < nrec + k.h.npb) {
    bwd_dh_item(a, k.h, (int)blockIdx.x - nrec);
  } else {
    if (threadIdx.x >= 256) return;   // the GEMM role is four waves; the others retire (a barrier counts live waves only)
    gemm_dw_body(g, gsm, blockIdx.x - (unsigned)nrec - (PROD ? (unsigned)k.h.npb : 0u));   // the monitor, then one item per workgroup in dispatch order
  }
}

}  // namespace clstm
