"""Time the contraction-major bf16 GEMM (clstm_debug_gemm mode 32) at the configs[4] weight-gradient shape.
usage: python scripts/gpu_gemm_mc_time.py [variant ...]   (CLSTM_HIP_VARIANT libraries; '' = the product library)
Each variant runs in its own process (the library is loaded once per process)."""
import os, subprocess, sys

CHILD = r'''
import os, sys, time, ctypes, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from clstm_amd.abi import load
lib = load()
R, Cn, K, ns = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
A = torch.randn(K, R, device="cuda").to(torch.bfloat16).contiguous()
B = torch.randn(K, Cn, device="cuda").to(torch.bfloat16).contiguous()
C = torch.zeros(R, Cn, device="cuda")
P = lambda t: ctypes.c_void_p(t.data_ptr())
def run(): lib.call("clstm_debug_gemm", 32, P(A), P(B), P(C), R, Cn, K, ns)
for _ in range(3): run()
lib.call("clstm_synchronize")
t0 = time.perf_counter()
n = 20
for _ in range(n): run()
lib.call("clstm_synchronize")
dt = (time.perf_counter() - t0) / n
print("R %d Cn %d K %d ns %d: %.1f us  %.0f TFLOP/s" % (R, Cn, K, ns, dt * 1e6, 2.0 * R * Cn * K / dt / 1e12))
'''
shapes = [(1544, 2048, 25600, 2), (568, 2048, 25600, 3)]
if os.environ.get("GEMM_SHAPES"):   # "R,Cn,K,ns;R,Cn,K,ns"
    shapes = [tuple(int(x) for x in t.split(",")) for t in os.environ["GEMM_SHAPES"].split(";")]
for v in [""] + sys.argv[1:]:
    env = dict(os.environ, CLSTM_HIP_VARIANT=v)
    for sh in (shapes if os.environ.get("GEMM_SHAPES") else shapes[:1 if v else 2]):
        out = subprocess.run([sys.executable, "-c", CHILD] + [str(x) for x in sh], env=env, capture_output=True, text=True, timeout=300)
        print("[%s]" % (v or "base"), (out.stdout.strip().splitlines() or [out.stderr[-300:]])[-1], flush=True)
