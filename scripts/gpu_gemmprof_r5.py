"""Round 5 diagnostics (library built with -DCLSTM_GEMM_PROF: make -C clstm_amd/csrc variant VARIANT=gprof EXTRA=-DCLSTM_GEMM_PROF,
run with CLSTM_HIP_VARIANT=gprof): where a 32-k block of the staggered 256 x 256 bf16-source GEMM goes -- shader-clock sums of
five segments of waves 0 (group A) and 4 (group B) of workgroup 0, and the kernel's duration with parts of the loop LEFT OUT
(results wrong, timing informative): 1 = no MFMAs, 2 = no global loads, 4 = no LDS staging writes, 8 = no fragment reads."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from clstm_amd.abi import load
lib = load()
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); lib.call("clstm_set_stream", stream.cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
SEG = ["staged (wait for the block's loads + ds_write)", "loads + fragment reads issued", "fragments arrived + barrier", "MFMAs issued", "barrier"]
for label, R, Cn, K in [("cube 4096^3", 4096, 4096, 4096), ("x.d L2 25600x1024x4096", 25600, 1024, 4096), ("W_x L2 25600x4096x1024", 25600, 4096, 1024)]:
    A = torch.randn(R, K, device="cuda").to(torch.bfloat16).contiguous()
    B = torch.randn(Cn, K, device="cuda").to(torch.bfloat16).contiguous()
    C = torch.zeros(R, Cn, device="cuda")
    print("==", label, flush=True)
    for lo in (0, 1, 2, 4, 8, 12, 14, 0):
        for _ in range(2):
            lib.call("clstm_debug_gemm", 30, P(A), P(B), P(C), R, Cn, K, lo)
        torch.cuda.synchronize()
        ts = []
        for r in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(5):
                lib.call("clstm_debug_gemm", 30, P(A), P(B), P(C), R, Cn, K, lo)
            e1.record(stream); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 5 * 1e3)
        out = (ctypes.c_longlong * 16)()
        rc = lib.dll.clstm_debug_gemm_prof(out)
        nkb = (K + 95) // 96 * 3
        seg = "  ".join("%s[A %.0f | B %.0f]" % (i, out[i] / nkb, out[8 + i] / nkb) for i in range(5))
        print("leave-out %2d: median %.1f us (min %.1f) | cycles per block, segments 0-4 %s | sum A %.0f B %.0f" %
              (lo, float(np.median(ts)), min(ts), seg, sum(out[:5]) / nkb, sum(out[8:13]) / nkb), flush=True)
print("segments:", "; ".join("%d = %s" % (i, s) for i, s in enumerate(SEG)))
