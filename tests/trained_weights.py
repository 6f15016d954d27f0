"""The "trained-like" weight set of SURVEY.md section 8(d): the uw3 architecture (48 -> BiLSTM(100) -> 83 classes) after 500
steps of the ORACLE's online SGD on the reference's OCR fixture (misc/textline.bin.png, "performance analysis"), so that gates
are not all ~0.5 and the posteriors are peaked -- plus real-line inputs for it: crops of the normalised fixture with jitter.

Test infrastructure (it drives oracle/): only tests/ imports this.  Nothing generated here is committed -- the generator is
(VERDICT r5 item 1a); 500 steps take the oracle ~10 s, cached per process."""
import numpy as np

from clstm_amd.fixture import GT, fixture_frames, fixture_transcript, jittered_crops  # noqa: F401

NI, NH, NC = 48, 100, 83
_cache = {}


def trained_like_params(ora32, steps=500, lr=1e-2, mom=0.9, seed=0.222):
    """parameters after `steps` online-SGD steps (CLSTMOCR::train, clstmhl.h:201-223) of the oracle on the fixture line"""
    key = (steps, lr, mom, seed)
    if key not in _cache:
        from clstm_amd.init import init_params
        from oracle.oracle import OracleNet
        x, tr = fixture_frames(), fixture_transcript()
        ref = OracleNet(ora32, NI, NH, NC, init=False)
        ref.set_params(init_params(NI, NH, NC, seed=seed))
        ref.set_lr(lr, mom)
        dec = None
        for _ in range(steps):
            dec = ref.train_line(x, tr)
        _cache[key] = (ref.get_params().copy(), np.asarray(dec).tolist() == tr.tolist())
    return _cache[key]


def fixture_crops(rng, T_list, ora_net=None):
    """real-line inputs (clstm_amd.fixture.jittered_crops).  With `ora_net` (an OracleNet holding the trained parameters) the
    transcript of a crop is what the oracle decodes on it (at least one label), i.e. a target the alignment is confident
    about; without, random labels."""
    lines, trs = jittered_crops(rng, T_list), []
    for w, T in zip(lines, T_list):
        if ora_net is not None:
            ora_net.set_inputs(w)
            ora_net.forward()
            d = np.asarray(ora_net.decode(), np.int32)
            trs.append(d if len(d) and 2 * len(d) + 1 <= T else np.array([1], np.int32))
        else:
            trs.append(rng.integers(1, 15, max(1, T // 8)).astype(np.int32))
    return lines, trs
