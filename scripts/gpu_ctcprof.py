import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from clstm_amd import abi
from clstm_amd.abi import ptr
lib = abi.load()
rng = np.random.default_rng(0)
for (T, L, nc, bs) in [(200, 25, 83, 64), (200, 25, 83, 1), (400, 80, 83, 16), (400, 50, 100, 64)]:   # (the last: configs[4])
    probs = rng.random((bs * T, nc)).astype(np.float32) ** 3
    probs /= probs.sum(1, keepdims=True)
    S = 2 * L + 1
    states = np.zeros(bs * S, np.int32); states[1::2] = 0
    st = []
    for b in range(bs):
        tr = rng.integers(1, nc, L); s = np.zeros(S, np.int32); s[1::2] = tr; st.append(s)
    states = np.concatenate(st)
    loff = (np.arange(bs + 1) * T).astype(np.int32); soff = (np.arange(bs + 1) * S).astype(np.int32)
    P = torch.from_numpy(probs).cuda(); D = torch.zeros_like(P)
    for it in range(3):
        lib.call("clstm_ctc_align_batch", ptr(P), ptr(D), None, nc, ptr(loff), ptr(states), ptr(soff), bs)
    cyc = np.zeros(16, np.int64)
    lib.call("clstm_debug_ctc_cycles", ptr(cyc))
    d = np.diff(cyc[:6])
    print("T=%d S=%d bs=%d phases A,B,C,D,E cycles:" % (T, S, bs), d.tolist(), "total", int(cyc[5]-cyc[0]))
    if cyc[6] > cyc[2]:   # short-line path: sub-phase stamps
        seq = [cyc[0], cyc[12], cyc[13], cyc[14], cyc[15], cyc[1], cyc[2], cyc[6], cyc[7], cyc[3], cyc[8], cyc[9], cyc[4], cyc[10], cyc[11], cyc[5]]
        names = ["classify", "A tile", "A row sums", "A log", "A rows->HBM", "B", "C load+max", "C max bcast", "C limexp", "D sums", "D totals",
                 "D normalise", "E project/blank", "E frame totals", "E write-out"]
        print("   " + ", ".join("%s %d" % (n, int(b - a)) for n, a, b in zip(names, seq[:-1], seq[1:])))
    else:   # tiled path: the parts of phase E, summed over the tiles
        print("   E: stage lattice tile + clear rows %d, project states onto classes %d, frame totals %d, write-out %d" % tuple(int(x) for x in cyc[6:10]))
