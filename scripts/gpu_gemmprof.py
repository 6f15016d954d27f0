import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from clstm_amd import abi
from clstm_amd.abi import ptr
lib = abi.load()
def run(mode, R, Cn, K, ns, reps=10):
    A = torch.randn(K, R, device="cuda") if mode >= 2 else torch.randn(R, K, device="cuda")
    B = torch.randn(K, Cn, device="cuda") if mode != 1 else torch.randn(Cn, K, device="cuda")
    C = torch.zeros(R, Cn, device="cuda")
    for _ in range(3): lib.call("clstm_debug_gemm", mode, ptr(A), ptr(B), ptr(C), R, Cn, K, ns)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): lib.call("clstm_debug_gemm", mode, ptr(A), ptr(B), ptr(C), R, Cn, K, ns)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("mode %d R=%d Cn=%d K=%d ns=%d: %.1f us  %.1f TFLOP/s" % (mode, R, Cn, K, ns, ms * 1e3, 2.0 * R * Cn * K / ms / 1e9))
for ns in (1, 8, 32, 72, 128):
    run(3, 149, 400, 12800, ns)
for ns in (8, 32, 72):
    run(2, 149, 400, 12800, ns)
run(3, 201, 83, 12800, 128); run(3, 201, 83, 12800, 32); run(2, 201, 83, 12800, 32)
run(3, 160, 512, 51200, 64); run(3, 160, 128, 12800, 1)
run(0, 12800, 800, 48, 1); run(0, 12800, 83, 200, 1); run(1, 12800, 200, 83, 1)
