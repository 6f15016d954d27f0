"""Host mirror of the reference's weight initialisation (batches.cc:11-52, clstm.cc:30-36).

`rinit` draws from the 48-bit-free LCG  state = frac(189843.9384938*state + 0.328340981343)
(double arithmetic, seed from $seed or 0.1) in row-major (i outer, j inner) order into
column-major Params; layers initialise in construction order: forward NPLSTM (WGI, WGF, WGO,
WCI), reversed NPLSTM, ..., SoftmaxLayer (clstm.cc:587-590, clstm_prefab.cc:52-68), while the flat
buffer is in walk_params order (WCI, WGF, WGI, WGO per NPLSTM; clstm.cc:59-62).
"""
import math
import os

import numpy as np


class LCG:
    def __init__(self, seed=None):
        if seed is None:
            seed = float(os.environ["seed"]) if "seed" in os.environ else 0.1
        self.state = float(seed)

    def randu(self):
        s = 189843.9384938 * self.state + 0.328340981343    # two roundings, as the C++ does
        self.state = s - math.floor(s)
        return self.state

    def rinit(self, n, m, s=0.01, mode="negbiased", offset=0.0):
        """Returns the column-major (n x m) block as a flat float32 array."""
        s = float(np.float32(s))
        a = np.zeros((n, m), np.float32)
        for i in range(n):
            for j in range(m):
                u = self.randu()
                if mode == "negbiased":
                    a[i, j] = 3 * s * u - 2 * s + offset
                elif mode == "unif":
                    a[i, j] = 2 * s * u - s + offset
                elif mode == "pos":
                    a[i, j] = s * u + offset
                elif mode == "neg":
                    a[i, j] = -s * u + offset
                else:
                    raise ValueError(mode)
        return a.T.reshape(-1).copy()       # column-major storage


def init_params(ninput, nhidden, nclasses, unidirectional=False, seed=None, scale=0.01,
                mode="negbiased"):
    """Flat parameter vector of make_net("bidi"/"bidi2"/"lstm1") after initialize()."""
    lcg = LCG(seed)
    nhidden = list(nhidden) if isinstance(nhidden, (list, tuple)) else [nhidden]
    blocks = []
    ni = ninput
    for no in nhidden:
        for _ in range(1 if unidirectional else 2):
            w = {}
            for name in ("WGI", "WGF", "WGO", "WCI"):          # init order
                w[name] = lcg.rinit(no, ni + no + 1, scale, mode)
            for name in ("WCI", "WGF", "WGI", "WGO"):          # flat (alphabetical) order
                blocks.append(w[name])
        ni = (1 if unidirectional else 2) * no
    blocks.append(lcg.rinit(nclasses, ni + 1, scale, mode))
    return np.concatenate(blocks)
