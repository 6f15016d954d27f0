// lstm_seq.h -- persistent per-line NPLSTM recurrence kernels (forward and backward).
//
// Replaces the per-timestep op chain of GenericNPLSTM::forward / ::backward
// (clstm.cc:600-653: forward_stack_delay, 4x forward_full1, forward_statemem,
// forward_nonlingate and their backward counterparts, clstm_compute.cc:275-547) by ONE
// launch per layer and pass: workgroup (line b, direction d) walks its own T_b frames without
// returning to the host.  `Reversed` (clstm.cc:458-478) is index arithmetic: direction 1
// visits frame T-1-s at its own step s.
//
// Only the recurrent half R.h_{t-1} of the gate products is inside the loop; the input half
// W_x.x_t + b for every frame of the minibatch is one hoisted MFMA GEMM (gemm_mfma.h), whose
// result G[token][dir][cell][gate] this kernel overwrites in place with the activations.
//
// Thread layout ("wavefront per cell tile"): wave w owns cells 16w..16w+15.
//   forward : lane = 4*cell_local + q.  Lane (cell,q) keeps the 4 gates x KQP recurrent weights
//             R_g[cell][k], k in quarter q, in VGPRs for the whole sequence; per step it reads
//             its quarter of h_{t-1} from LDS (float4, conflict-free: quarter stride/4 is odd),
//             does 4*KQP FMAs, sums the four quarters with two DPP quad_perm adds, then lane q
//             applies gate q's nonlinearity, the quad exchanges the four activations by DPP
//             broadcast and every lane forms c_t and h_t (fused statemem + nonlingate).
//   backward: matvec phase lane = 16*r + js : k-group (4 output cells) x 16 slices of the 4*no
//             (gate,j) delta pairs, weights R_g[j][k] in VGPRs, deltas from LDS; a 4-step DPP
//             row_ror all-reduce leaves dh_rec for the 4 cells in every lane of the row; the
//             element-wise phase re-uses the forward's (cell, gate) = (lane>>2, lane&3) mapping.
// One __syncthreads per step (h / delta vector double-buffered in LDS).
#pragma once
#include "devintrin.h"

namespace clstm {

struct LstmSeqArgs {
  const float* Rpk;     // packed recurrent weights for this pass: [dir][16*NK4][nthreads]
  float* G;             // [N][2][4*no]  fwd: in pre-activation (x part + bias), out activation
  float* C;             // [N][2][no]    cell state
  float* H;             // [N][2*no]     outputs (dir d at column d*no) = Parallel's stacked output
  const float* dH;      // [N][2*no]     bwd: delta on H            (backward only)
  float* D;             // [N][2][4*no]  bwd: gate pre-activation deltas (backward only)
  const int* line_off;  // [bs+1] first token of each line
  int no;
  int ndir;             // 2 (bidirectional) or 1 (forward only, "lstm1")
};

constexpr int lstm_qstride(int nk4) { return 4 * nk4 + ((nk4 & 1) ? 0 : 4); }

template <int NK4>
__global__ __launch_bounds__(64 * NK4) void lstm_fwd_kernel(LstmSeqArgs a) {
  constexpr int KQP = 4 * NK4;
  constexpr int QS = KQP + ((NK4 & 1) ? 0 : 4);
  constexpr int HB = 4 * QS;
  float* lds = dyn_smem<float>();  // hbuf[2][HB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nthreads = blockDim.x;
  const int b = blockIdx.x, dir = blockIdx.y;
  const int no = a.no;
  const int q = lane & 3, cell = wave * 16 + (lane >> 2);
  const bool valid = cell < no;
  const int nd = a.ndir;

  float w[4][KQP];
  {
    const float* rp = a.Rpk + (size_t)dir * 4 * KQP * nthreads + tid;
#pragma unroll
    for (int g = 0; g < 4; g++)
#pragma unroll
      for (int kk = 0; kk < KQP; kk++) w[g][kk] = rp[(size_t)(g * KQP + kk) * nthreads];
  }
  for (int i = tid; i < 2 * HB; i += nthreads) lds[i] = 0.0f;

  const int off = a.line_off[b];
  const int T = a.line_off[b + 1] - off;
  const int hslot = (cell / KQP) * QS + (cell % KQP);
  float c_prev = 0.0f;
  const size_t gstride = (size_t)nd * 4 * no;
  const size_t gofs = (size_t)dir * 4 * no + cell * 4 + q;
  auto tok = [&](int t) -> size_t { return (size_t)(dir == 0 ? off + t : off + T - 1 - t); };
  float gx = (valid && T > 0) ? a.G[tok(0) * gstride + gofs] : 0.0f;
  __syncthreads();
  for (int t = 0; t < T; t++) {
    const size_t tk = tok(t);
    const float gx_next = (valid && t + 1 < T) ? a.G[tok(t + 1) * gstride + gofs] : 0.0f;
    const float* hq = lds + (t & 1) * HB + q * QS;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
#pragma unroll
    for (int j = 0; j < NK4; j++) {
      const float4 hv = *reinterpret_cast<const float4*>(hq + 4 * j);
      acc0 += w[0][4 * j] * hv.x; acc1 += w[1][4 * j] * hv.x; acc2 += w[2][4 * j] * hv.x; acc3 += w[3][4 * j] * hv.x;
      acc0 += w[0][4 * j + 1] * hv.y; acc1 += w[1][4 * j + 1] * hv.y; acc2 += w[2][4 * j + 1] * hv.y; acc3 += w[3][4 * j + 1] * hv.y;
      acc0 += w[0][4 * j + 2] * hv.z; acc1 += w[1][4 * j + 2] * hv.z; acc2 += w[2][4 * j + 2] * hv.z; acc3 += w[3][4 * j + 2] * hv.z;
      acc0 += w[0][4 * j + 3] * hv.w; acc1 += w[1][4 * j + 3] * hv.w; acc2 += w[2][4 * j + 3] * hv.w; acc3 += w[3][4 * j + 3] * hv.w;
    }
    // sum the four k-quarters held by the quad
    acc0 += quad_xor1(acc0); acc1 += quad_xor1(acc1); acc2 += quad_xor1(acc2); acc3 += quad_xor1(acc3);
    acc0 += quad_xor2(acc0); acc1 += quad_xor2(acc1); acc2 += quad_xor2(acc2); acc3 += quad_xor2(acc3);
    // lane q finishes gate q: q=0 gi, 1 gf, 2 go (sigmoid); 3 ci (tanh)   [forward_full1]
    const float pre = (q == 0 ? acc0 : q == 1 ? acc1 : q == 2 ? acc2 : acc3) + gx;
    const float act = (q == 3) ? tanh_dev(pre) : sigmoid_dev(pre);
    const float gi = quad_bcast<0>(act), gf = quad_bcast<1>(act), go = quad_bcast<2>(act),
                ci = quad_bcast<3>(act);
    float c = ci * gi;                 // forward_statemem (clstm_compute.cc:504-508)
    if (t > 0) c += gf * c_prev;
    const float h = tanh_dev(c) * go;  // forward_nonlingate (clstm_compute.cc:530-537)
    c_prev = c;
    if (valid) {
      a.G[tk * gstride + gofs] = act;
      if (q == 0) {
        a.C[(tk * nd + dir) * no + cell] = c;
        a.H[tk * nd * no + (size_t)dir * no + cell] = h;
        lds[((t + 1) & 1) * HB + hslot] = h;
      }
    }
    gx = gx_next;
    __syncthreads();
  }
}

template <int NK4>
__global__ __launch_bounds__(64 * NK4) void lstm_bwd_kernel(LstmSeqArgs a) {
  constexpr int SLP = 4 * NK4;
  constexpr int QS = SLP + ((NK4 & 1) ? 0 : 4);
  constexpr int DB = 16 * QS;
  float* lds = dyn_smem<float>();  // dbuf[2][DB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nthreads = blockDim.x;
  const int b = blockIdx.x, dir = blockIdx.y;
  const int no = a.no, nd = a.ndir;
  const int SL = (4 * no + 15) / 16;  // (gate,j) pairs per slice
  const int js = lane & 15;
  const int g = lane & 3, cell = wave * 16 + (lane >> 2), isel = (lane >> 2) & 3;
  const bool valid = cell < no;

  float wb[4][SLP];
  {
    const float* rp = a.Rpk + (size_t)dir * 4 * SLP * nthreads + tid;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int pp = 0; pp < SLP; pp++) wb[i][pp] = rp[(size_t)(i * SLP + pp) * nthreads];
  }
  for (int i = tid; i < 2 * DB; i += nthreads) lds[i] = 0.0f;

  const int off = a.line_off[b];
  const int T = a.line_off[b + 1] - off;
  const int pidx = g * no + cell;  // this lane's delta goes to pair (gate g, j = cell)
  const int dslot = (pidx / SL) * QS + (pidx % SL);
  const size_t gstride = (size_t)nd * 4 * no;
  const size_t gofs = (size_t)dir * 4 * no + cell * 4 + g;
  auto tok = [&](int s) -> size_t { return (size_t)(dir == 0 ? off + s : off + T - 1 - s); };

  // prefetch for step s = T-1
  float act = 0.f, dhv = 0.f, c_cur = 0.f, c_m1 = 0.f;
  if (valid && T > 0) {
    const size_t tk = tok(T - 1);
    act = a.G[tk * gstride + gofs];
    dhv = a.dH[tk * nd * no + (size_t)dir * no + cell];
    c_cur = a.C[(tk * nd + dir) * no + cell];
    if (T > 1) c_m1 = a.C[(tok(T - 2) * nd + dir) * no + cell];
  }
  float dc_carry = 0.0f;
  __syncthreads();
  int cur = 0;
  for (int s = T - 1; s >= 0; s--) {
    const size_t tk = tok(s);
    // prefetch step s-1 (and c of step s-2)
    float act_n = 0.f, dhv_n = 0.f, c_m2 = 0.f;
    if (valid && s > 0) {
      const size_t tn = tok(s - 1);
      act_n = a.G[tn * gstride + gofs];
      dhv_n = a.dH[tn * nd * no + (size_t)dir * no + cell];
      if (s > 1) c_m2 = a.C[(tok(s - 2) * nd + dir) * no + cell];
    }
    // dh_rec[k] = sum_{g,j} R_g[j][k] * delta_g[j](s+1)      [backward_lin1 recurrent half +
    //                                                         backward_stack_delay, :294-304,:398-410]
    const float* dq = lds + cur * DB + js * QS;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
#pragma unroll
    for (int j = 0; j < NK4; j++) {
      const float4 dv = *reinterpret_cast<const float4*>(dq + 4 * j);
      acc0 += wb[0][4 * j] * dv.x; acc1 += wb[1][4 * j] * dv.x; acc2 += wb[2][4 * j] * dv.x; acc3 += wb[3][4 * j] * dv.x;
      acc0 += wb[0][4 * j + 1] * dv.y; acc1 += wb[1][4 * j + 1] * dv.y; acc2 += wb[2][4 * j + 1] * dv.y; acc3 += wb[3][4 * j + 1] * dv.y;
      acc0 += wb[0][4 * j + 2] * dv.z; acc1 += wb[1][4 * j + 2] * dv.z; acc2 += wb[2][4 * j + 2] * dv.z; acc3 += wb[3][4 * j + 2] * dv.z;
      acc0 += wb[0][4 * j + 3] * dv.w; acc1 += wb[1][4 * j + 3] * dv.w; acc2 += wb[2][4 * j + 3] * dv.w; acc3 += wb[3][4 * j + 3] * dv.w;
    }
    // all-reduce over the 16 slices of the row
    acc0 += row_ror<8>(acc0); acc1 += row_ror<8>(acc1); acc2 += row_ror<8>(acc2); acc3 += row_ror<8>(acc3);
    acc0 += row_ror<4>(acc0); acc1 += row_ror<4>(acc1); acc2 += row_ror<4>(acc2); acc3 += row_ror<4>(acc3);
    acc0 += row_ror<2>(acc0); acc1 += row_ror<2>(acc1); acc2 += row_ror<2>(acc2); acc3 += row_ror<2>(acc3);
    acc0 += row_ror<1>(acc0); acc1 += row_ror<1>(acc1); acc2 += row_ror<1>(acc2); acc3 += row_ror<1>(acc3);
    const float dh_rec = isel == 0 ? acc0 : isel == 1 ? acc1 : isel == 2 ? acc2 : acc3;

    const float gi = quad_bcast<0>(act), gf = quad_bcast<1>(act), go = quad_bcast<2>(act),
                ci = quad_bcast<3>(act);
    const float dh = dhv + dh_rec;             // out[s].d, clstm.cc:626-628 + :646
    const float th = tanh_dev(c_cur);          // backward_nonlingate recomputes tanh(state)
    const float d_go = th * dh;                //   go.d += t * out.d
    const float dc = dc_carry + (-th * th + 1.0f) * (go * dh);  // state.d += (1-t^2) * (go*out.d)
    float d_gf = 0.0f;
    if (s > 0) {                               // backward_statemem (clstm_compute.cc:509-515)
      dc_carry = dc * gf;
      d_gf = dc * c_m1;
    }
    const float d_gi = dc * ci, d_ci = dc * gi;
    // backward_nonlin0 in place (clstm_compute.cc:231-267): y(1-y) for SIG, 1-y^2 for TANH
    float delta;
    if (g == 0) delta = gi * (-gi + 1.0f) * d_gi;
    else if (g == 1) delta = gf * (-gf + 1.0f) * d_gf;
    else if (g == 2) delta = go * (-go + 1.0f) * d_go;
    else delta = (-ci * ci + 1.0f) * d_ci;
    if (valid) {
      a.D[tk * gstride + gofs] = delta;
      lds[(cur ^ 1) * DB + dslot] = delta;
    }
    act = act_n; dhv = dhv_n; c_cur = c_m1; c_m1 = c_m2;
    cur ^= 1;
    __syncthreads();
  }
}

}  // namespace clstm
