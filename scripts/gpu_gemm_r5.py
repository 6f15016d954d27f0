"""Round 5: times the bf16-source GEMMs (clstm_debug_gemm modes 34 / 30 / 31 = kk LDS-DMA / staggered / one-barrier, 32 / 33 = mc staggered /
one-barrier) at the configs[4] shapes, every variant interleaved in ONE process (HIP events on the library stream, R rounds,
median and min).  usage: python scripts/gpu_gemm_r5.py [rounds]"""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from clstm_amd.abi import load
lib = load()
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); lib.call("clstm_set_stream", stream.cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 7
# (label, modes, R, Cn, K, nsplit): kk: A [R][K], B [Cn][K]; mc: A [K][R], B [K][Cn]
CASES = [("W_x L2  25600x4096x1024 (kk)", (34, 30, 31), 25600, 4096, 1024, 1),
         ("x.d L2  25600x1024x4096 (kk)", (34, 30, 31), 25600, 1024, 4096, 1),
         ("W_x L1  25600x4096x64   (kk)", (34, 30, 31), 25600, 4096, 64, 1),
         ("cube    4096^3          (kk)", (34, 30, 31), 4096, 4096, 4096, 1),
         ("cube    8192^3          (kk)", (34, 30), 8192, 8192, 8192, 1),
         ("dW L2   1544x2048x25600 ns2 (mc)", (35, 32, 33), 1544, 2048, 25600, 2),
         ("dW L2   1536x2048x25600 ns2 (mc)", (35, 32, 33), 1536, 2048, 25600, 2),
         ("dW L2   1536x2048x25600 ns5 (mc)", (35, 32, 33), 1536, 2048, 25600, 5),
         ("dW L2   1528x2048x25600 ns5 (mc)", (35, 32), 1528, 2048, 25600, 5),
         ("dW L1   584x2048x25600 ns3 (mc)", (35, 32, 33), 584, 2048, 25600, 3),
         ("dW L1   576x2048x25600 ns5 (mc)", (35, 32, 33), 576, 2048, 25600, 5)]
for label, modes, R, Cn, K, ns in CASES:
    kk = modes[0] in (30, 31)
    A = torch.randn((R, K) if kk else (K, R), device="cuda").to(torch.bfloat16).contiguous()
    B = torch.randn((Cn, K) if kk else (K, Cn), device="cuda").to(torch.bfloat16).contiguous()
    C = {m: torch.zeros(R, Cn, device="cuda") for m in modes}
    times = {m: [] for m in modes}
    for m in modes:
        for _ in range(2):
            lib.call("clstm_debug_gemm", m, P(A), P(B), P(C[m]), R, Cn, K, ns)
    torch.cuda.synchronize()
    for r in range(ROUNDS):
        for m in modes:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(5):
                lib.call("clstm_debug_gemm", m, P(A), P(B), P(C[m]), R, Cn, K, ns)
            e1.record(stream)
            torch.cuda.synchronize()
            times[m].append(e0.elapsed_time(e1) / 5 * 1e3)
    same = all(torch.equal(C[modes[0]], C[m]) for m in modes[1:])
    fl = 2.0 * R * Cn * K
    print(label, "| bit-identical:", same, "|", "  ".join("mode %d: median %.1f us (min %.1f) = %.0f TF/s" % (m, float(np.median(times[m])), min(times[m]), fl / float(np.median(times[m])) / 1e6) for m in modes), flush=True)
    del A, B, C
