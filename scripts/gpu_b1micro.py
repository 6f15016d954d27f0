"""Round 5: the small launches of the headline step in isolation -- the softmax layer's x.d product (clstm_debug_gemm modes 21 / 24:
64 x 64 and 128 x 128 tiles of the f32-grade kernel) at 12800 x 200 x 83."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from clstm_amd.abi import load
lib = load()
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); lib.call("clstm_set_stream", stream.cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
for label, modes, R, Cn, K in [("x.d B1 12800x200x83", (21, 24), 12800, 200, 83), ("x.d mb256 51200x200x83", (21, 24), 51200, 200, 83)]:
    A = torch.randn(R, K, device="cuda"); B = torch.randn(Cn, K, device="cuda")
    C = {m: torch.zeros(R, Cn, device="cuda") for m in modes}
    times = {m: [] for m in modes}
    for m in modes:
        for _ in range(3): lib.call("clstm_debug_gemm", m, P(A), P(B), P(C[m]), R, Cn, K, 1)
    torch.cuda.synchronize()
    for r in range(9):
        for m in modes:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(20): lib.call("clstm_debug_gemm", m, P(A), P(B), P(C[m]), R, Cn, K, 1)
            e1.record(stream); torch.cuda.synchronize()
            times[m].append(e0.elapsed_time(e1) / 20 * 1e3)
    ref = A.double() @ B.double().T
    print(label, "|", "  ".join("mode %d: median %.2f us (min %.2f) err %.1e" % (m, float(np.median(times[m])), min(times[m]), float((C[m].double() - ref).abs().max())) for m in modes), flush=True)
