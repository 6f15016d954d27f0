"""Pins the hardware facts the kernels rely on: DPP lane-move encodings, the f32 MFMA
fragment layout (through the GEMM with asymmetric operands, cdna guide §3 "A=I-check with
ASYMMETRIC B") and the device activation functions.  'emu' checks the emulator's model of them,
'hip' (-m gpu) checks the silicon -- both must agree with the same expectations."""
import numpy as np
import pytest

from clstm_amd.abi import ptr
from common import assert_close


def test_lane_ops(backend):
    out = backend.zeros((11, 64))
    backend.lib.call("clstm_debug_lane_ops", ptr(out))
    got = backend.down(out)
    lane = np.arange(64)
    want = np.stack([lane ^ 1, lane ^ 2, (lane & ~3) | 0, (lane & ~3) | 1, (lane & ~3) | 2, (lane & ~3) | 3,
                     (lane & ~15) | ((lane + 1) & 15), (lane & ~15) | ((lane + 4) & 15),
                     (lane & ~15) | ((lane + 8) & 15)]).astype(np.float32)
    # row_ror all-reduce only needs SOME rotation by N within the row: accept either direction
    for k in range(6):
        assert np.array_equal(got[k], want[k]), k
    assert np.array_equal(got[9], (lane ^ 7).astype(np.float32))            # row_half_mirror
    assert np.array_equal(got[10][1:], lane[:-1].astype(np.float32))        # wave_shr:1 (lane 0 unspecified)
    for k, n in ((6, 1), (7, 4), (8, 8)):
        alt = ((lane & ~15) | ((lane - n) & 15)).astype(np.float32)
        assert np.array_equal(got[k], want[k]) or np.array_equal(got[k], alt), k


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("R,Cn,K,ns", [(64, 64, 16, 1), (70, 45, 37, 1), (149, 400, 333, 5), (5, 3, 2, 1),
                                       (130, 83, 200, 3)])
def test_gemm_modes(backend, mode, R, Cn, K, ns):
    rng = np.random.default_rng(R * 1000 + Cn)
    A = rng.normal(size=(R, K)).astype(np.float32)
    B = rng.normal(size=(K, Cn)).astype(np.float32)
    want = A.astype(np.float64) @ B.astype(np.float64)
    Ad = backend.up(A if mode < 2 else A.T)
    Bd = backend.up(B if mode != 1 else B.T)
    Cd = backend.zeros((R, Cn))
    backend.lib.call("clstm_debug_gemm", mode, ptr(Ad), ptr(Bd), ptr(Cd), R, Cn, K, ns if mode >= 2 else 1)
    got = backend.down(Cd)
    scale = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
    assert (np.abs(got - want) <= 2e-6 * scale + 1e-6).all()


@pytest.mark.parametrize("mode", [20, 21, 22])
@pytest.mark.parametrize("R,Cn,K,ns", [(64, 64, 32, 1), (70, 45, 37, 1), (149, 400, 333, 5), (5, 3, 2, 1), (130, 83, 200, 3),
                                       (200, 83, 49, 1), (97, 200, 1000, 2)])
def test_gemm_x3_modes(backend, mode, R, Cn, K, ns):
    """f32-grade GEMM on the bf16 MFMA (operands split hi + lo, three products; gemm_bf16.h gemm_x3_body): against the
    float64 product of the f32 inputs, within 1e-5 of sum |a||b| (the f32 MFMA kernel is held to 2e-6)."""
    rng = np.random.default_rng(R * 1000 + Cn + 3)
    A = rng.normal(size=(R, K)).astype(np.float32)
    B = rng.normal(size=(K, Cn)).astype(np.float32)
    want = A.astype(np.float64) @ B.astype(np.float64)
    Ad = backend.up(A if mode < 22 else A.T)
    Bd = backend.up(B if mode != 21 else B.T)
    Cd = backend.zeros((R, Cn))
    backend.lib.call("clstm_debug_gemm", mode, ptr(Ad), ptr(Bd), ptr(Cd), R, Cn, K, ns if mode >= 22 else 1)
    got = backend.down(Cd)
    scale = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
    err = np.abs(got - want) / (scale + 1e-30)
    assert (np.abs(got - want) <= 1e-5 * scale + 1e-6).all(), float(err.max())


@pytest.mark.parametrize("mode", [23, 24, 25])
@pytest.mark.parametrize("R,Cn,K,ns", [(128, 128, 32, 1), (256, 256, 64, 1), (97, 200, 31, 1), (300, 130, 1000, 2), (577, 260, 129, 3)])
def test_gemm_x3_big_tile_modes(backend, mode, R, Cn, K, ns):
    """the same f32-grade product on 128 x 128 tiles (gemm_x3_128_kernel: the backward products of wide layers in the exact-f32
    mode): whole tiles, ragged edges, k tails, split-K slabs; asymmetric random operands, so a transposed fragment or a
    swapped hi / lo image would show.  Same bar as the 64 x 64 kernel: 1e-5 of sum |a||b| against the float64 product."""
    rng = np.random.default_rng(R * 1000 + Cn + 11)
    A = rng.normal(size=(R, K)).astype(np.float32)
    B = rng.normal(size=(K, Cn)).astype(np.float32)
    want = A.astype(np.float64) @ B.astype(np.float64)
    Ad = backend.up(A if mode < 25 else A.T)
    Bd = backend.up(B if mode != 24 else B.T)
    Cd = backend.zeros((R, Cn))
    backend.lib.call("clstm_debug_gemm", mode, ptr(Ad), ptr(Bd), ptr(Cd), R, Cn, K, ns if mode == 25 else 1)
    got = backend.down(Cd)
    scale = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
    err = np.abs(got - want) / (scale + 1e-30)
    assert (np.abs(got - want) <= 1e-5 * scale + 1e-6).all(), float(err.max())
    # (a plain bf16 product of the same operands sits at ~4e-3 of that scale: a swapped or missing lo image would show)
    assert float(err.max()) < 1.5e-5, float(err.max())


@pytest.mark.parametrize("mode", [10, 11, 12])
@pytest.mark.parametrize("R,Cn,K,ns", [(64, 64, 32, 1), (70, 45, 37, 1), (149, 400, 333, 5), (5, 3, 2, 1),
                                       (130, 83, 200, 3),
                                       # the 128 x 128-tile kernel (R, Cn >= 96): whole tiles, ragged edges, k tails, slabs
                                       (128, 128, 32, 1), (256, 256, 64, 1), (97, 200, 31, 1), (300, 130, 1000, 2),
                                       (577, 260, 129, 3)])
def test_gemm_bf16_modes(backend, mode, R, Cn, K, ns):
    """bf16-input / f32-accumulate GEMM (gemm_bf16.h): exact against a float64 product of the SAME inputs
    rounded to bf16 (asymmetric random operands, so a transposed fragment would show)."""
    rng = np.random.default_rng(R * 1000 + Cn + 7)
    A = rng.normal(size=(R, K)).astype(np.float32)
    B = rng.normal(size=(K, Cn)).astype(np.float32)

    def bf16(x):   # round to nearest even on the upper 16 bits
        u = x.view(np.uint32).astype(np.uint64)
        u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
        return u.astype(np.uint32).view(np.float32)
    want = bf16(A).astype(np.float64) @ bf16(B).astype(np.float64)
    m = mode - 10
    Ad = backend.up(A if m < 2 else A.T)
    Bd = backend.up(B if m != 1 else B.T)
    Cd = backend.zeros((R, Cn))
    backend.lib.call("clstm_debug_gemm", mode, ptr(Ad), ptr(Bd), ptr(Cd), R, Cn, K, ns if m >= 2 else 1)
    got = backend.down(Cd)
    scale = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
    assert (np.abs(got - want) <= 2e-6 * scale + 1e-6).all()


def test_device_activations(backend):
    # sigmoid_dev / tanh_dev (devintrin.h) through forward_nonlin0 over a wide sweep
    x = np.concatenate([np.linspace(-40, 40, 4001), np.linspace(-0.6, 0.6, 2001), [0.0, 1e-8, -1e-8, 1e-3]])
    x = x.astype(np.float32)
    for nl, f in ((1, lambda v: 1 / (1 + np.exp(-v))), (2, np.tanh)):
        y = backend.up(x)
        backend.lib.call("clstm_forward_nonlin0", ptr(y), x.size, nl)
        assert_close(backend.down(y), f(x.astype(np.float64)), rtol=1e-5, atol=1e-30, what="nl %d" % nl)


def _bf16_bits(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return u.astype(np.uint16)


@pytest.mark.parametrize("R,Cn,K", [(128, 128, 32), (256, 256, 64), (97, 200, 30), (300, 130, 1000), (129, 513, 96),
                                    (500, 512, 96), (250, 760, 70),    # (256, 256) and these two: 256 x 256 tiles
                                    (500, 760, 64), (250, 512, 416), (512, 256, 32)])   # LDS-DMA eligible (K % 32 == 0, K >= 64; the last: not)
def test_gemm_bf16_sources(backend, R, Cn, K):
    """128 x 128-tile GEMM whose operands are ALREADY bf16 and k-contiguous in memory (gemm_b16kk_128_kernel): exact
    against the float64 product of the same bf16 values (f32 accumulation: 2e-6 of sum |a||b|)."""
    rng = np.random.default_rng(R + Cn + K)
    A = rng.normal(size=(R, K)).astype(np.float32)
    B = rng.normal(size=(Cn, K)).astype(np.float32)
    Ab, Bb = _bf16_bits(A), _bf16_bits(B)
    Af = (Ab.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    Bf = (Bb.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    want = Af @ Bf.T
    Ad = backend.up(Ab, dtype=np.uint16)
    Bd = backend.up(Bb, dtype=np.uint16)
    Cd = backend.zeros((R, Cn))
    backend.lib.call("clstm_debug_gemm", 30, ptr(Ad), ptr(Bd), ptr(Cd), R, Cn, K, 1)
    got = backend.down(Cd)
    scale = np.abs(Af) @ np.abs(Bf).T
    assert (np.abs(got - want) <= 2e-6 * scale + 1e-6).all()
    # mode 31: the one-barrier loop; mode 30 runs the 256 x 256 tiles with the two waves of a SIMD half a block apart (two
    # barriers per block): same fragments, same MFMA order per accumulator -> bit-identical
    Ce = backend.zeros((R, Cn))
    backend.lib.call("clstm_debug_gemm", 31, ptr(Ad), ptr(Bd), ptr(Ce), R, Cn, K, 1)
    assert np.array_equal(backend.down(Ce), got)
    # mode 34: the operand tiles brought in by LDS-DMA (three buffers, nothing staged through registers; 256 x 256 tiles and
    # K % 32 == 0, else it is mode 30 again): same image, same fragments, same order
    Cf = backend.zeros((R, Cn))
    backend.lib.call("clstm_debug_gemm", 34, ptr(Ad), ptr(Bd), ptr(Cf), R, Cn, K, 1)
    assert np.array_equal(backend.down(Cf), got)


@pytest.mark.parametrize("R,Cn,K,ns", [(128, 128, 32, 1), (256, 256, 64, 1), (104, 200, 30, 1), (304, 136, 1000, 3), (136, 520, 700, 2),
                                       (504, 512, 700, 2), (248, 760, 130, 1),    # (256, 256) and these two: 256 x 256 tiles
                                       (576, 520, 130, 1), (380, 256, 230, 2)])   # 192 x 256 tiles (24 chunks per row: waves 6, 7 stage no A)
def test_gemm_bf16_contraction_major(backend, R, Cn, K, ns):
    """128 x 128-tile GEMM whose bf16 operands lie contraction-major in memory ([K][R], [K][Cn]: the weight-gradient
    product's frame-major deltas and sources), transposed by ds_read_b64_tr_b16 on the way into the MFMA
    (gemm_b16mc_128_kernel), with split-K slabs: exact against the float64 product of the same bf16 values.  The random
    operands make the check transpose-detecting."""
    rng = np.random.default_rng(R + Cn + K)
    A = rng.normal(size=(K, R)).astype(np.float32)
    B = rng.normal(size=(K, Cn)).astype(np.float32)
    Ab, Bb = _bf16_bits(A), _bf16_bits(B)
    Af = (Ab.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    Bf = (Bb.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    want = Af.T @ Bf
    Ad = backend.up(Ab, dtype=np.uint16)
    Bd = backend.up(Bb, dtype=np.uint16)
    Cd = backend.zeros((R, Cn))
    backend.lib.call("clstm_debug_gemm", 32, ptr(Ad), ptr(Bd), ptr(Cd), R, Cn, K, ns)
    got = backend.down(Cd)
    scale = np.abs(Af).T @ np.abs(Bf)
    assert (np.abs(got - want) <= 2e-6 * scale + 1e-6).all()
    Ce = backend.zeros((R, Cn))          # mode 33: the one-barrier loop (mode 32: staggered wave groups on 256 x 256 tiles)
    backend.lib.call("clstm_debug_gemm", 33, ptr(Ad), ptr(Bd), ptr(Ce), R, Cn, K, ns)
    assert np.array_equal(backend.down(Ce), got)
    Cf = backend.zeros((R, Cn))          # mode 35: the tiles brought in by LDS-DMA (big tiles; contraction tails: zeros from out-of-range requests)
    backend.lib.call("clstm_debug_gemm", 35, ptr(Ad), ptr(Bd), ptr(Cf), R, Cn, K, ns)
    assert np.array_equal(backend.down(Cf), got)

