#!/usr/bin/env python
"""Copy the judged summaries of one scripts/gpu_round.sh run from gpurun_out/<tag>/ into profiles/
(prefix r<NN>_) and rebuild profiles/pmc_r<NN>.json (HBM bytes per launch of the main kernels).

usage: python scripts/collect_profiles.py <tag> <NN>"""
import json, os, re, shutil, sys

tag, rnd = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
pre = "r%s_" % rnd
if os.path.exists(os.path.join(src, "bench_kernel_stats_strict.csv")):
    pass
for name in ("bench_default.json", "bench_mb1.json", "bench_mb16.json", "bench_mb256.json", "bench_mb1024.json", "bench_mb2048.json", "bench_trained.json",
             "bench_kernel_stats_strict.csv", "bench_kernel_stats_mb256.csv", "pmc_FETCH_SIZE_strict_summary.txt", "pmc_WRITE_SIZE_strict_summary.txt",
             "pmc_FETCH_SIZE_mb256_summary.txt", "pmc_WRITE_SIZE_mb256_summary.txt",
             "bench_ragged.json", "bench_b2.json", "bench_b2_bf16gemm.json", "bench_b2_bf16.json", "bench_forcedist.json",
             "bench_overlap0.json", "bench_host_inputs.json", "bench_driver_cmd.json", "fwd_timeline.txt", "lstm_fwd_phase_cycles.txt", "ctc_phase_cycles.txt", "xcd_phase_cycles.txt", "timeline.txt", "host.txt", "pmc_FETCH_SIZE_summary.txt", "pmc_WRITE_SIZE_summary.txt",
             "pmc_SQ_VALU_MFMA_BUSY_CYCLES_summary.txt", "pmc_GRBM_GUI_ACTIVE_summary.txt", "pmc_SQ_summary.txt",
             "pmc_FETCH_SIZE_ov0_summary.txt", "pmc_WRITE_SIZE_ov0_summary.txt", "b2_timeline.txt", "b2_kernel_stats.csv", "drop_in_rate.txt", "driver_rate.txt", "pytest_gpu.log"):
    p = os.path.join(src, name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, pre + name))
p = os.path.join(src, "prof", "bench_kernel_stats.csv")
if os.path.exists(p):
    shutil.copy(p, os.path.join(dst, pre + "bench_kernel_stats.csv"))


def summary(counter):
    out = {}
    if not os.path.exists(os.path.join(src, "pmc_%s_summary.txt" % counter)):
        return out
    for line in open(os.path.join(src, "pmc_%s_summary.txt" % counter)):
        m = re.match(r"(.*?)\s+launches\s+(\d+)\s+avg\s+([\d.]+)", line)
        if m:
            out[m.group(1).strip()] = float(m.group(3))
    return out


fe, wr = summary("FETCH_SIZE"), summary("WRITE_SIZE")
if os.path.exists(os.path.join(src, "pmc_FETCH_SIZE_ov0_summary.txt")):   # pure kernels (CLSTM_OVERLAP=0 passes): names do not clash
    for k, v in summary("FETCH_SIZE_ov0").items(): fe.setdefault(k, v)
    for k, v in summary("WRITE_SIZE_ov0").items(): wr.setdefault(k, v)
names = {"lstm_fwd": "void clstm::lstm_fwd_kernel<", "lstm_bwd": "void clstm::lstm_bwd_kernel<",
         "lstm_fwd_fused (W_x producers + recurrence + softmax consumers, one launch)": "void clstm::lstm_fwd_fused_kernel<",
         "lstm_bwd_dw (recurrence + weight-gradient GEMM, one launch)": "void clstm::lstm_bwd_dw_kernel<",
         "ctc_align": ("void clstm::ctc_align_kernel<false>", "clstm::ctc_align_kernel"), "sgd_update": "clstm::k_update",
         "gemm_dw (all split-K launches, avg)": "void clstm::gemm_f32_kernel<1, 1, clstm::StorePartial>",
         "gemm_gates_x / gemm_softmax (avg)": "void clstm::gemm_f32_kernel<0, 1, clstm::StoreBias>"}
kern = {}
def find(table, prefixes):
    for prefix in (prefixes if isinstance(prefixes, tuple) else (prefixes,)):
        for name, v in table.items():
            if name.startswith(prefix):
                return v
    return None


def leg_kernels(fe, wr):
    out = {}
    for k, full in names.items():
        f, w = find(fe, full), find(wr, full)
        if f is not None and w is not None:
            out[k] = {"fetch_kib": f, "write_kib": w, "hbm_bytes": int(round((2.0 * f + w) * 1024))}
    return out


kern = leg_kernels(fe, wr)
legs = {}
for leg, what in (("strict", "bench.py --strict-f32"), ("mb256", "bench.py --minibatch 256")):
    lk = leg_kernels(summary("FETCH_SIZE_" + leg), summary("WRITE_SIZE_" + leg))
    if lk:
        legs[leg] = {"command": what, "kernels": lk}
doc = {
    "_comment": "HBM traffic per launch from rocprofv3 PMC passes (profiles/%spmc_*_summary.txt), bench.py default "
                "workload (minibatch 64, T=200, BiLSTM(100)). FETCH_SIZE/WRITE_SIZE are KiB; on gfx950 FETCH_SIZE "
                "counts 128-B requests at 64 B (MI355X_MICROARCH.md, HBM) -> reads are doubled; WRITE_SIZE is exact. "
                "Calibration in the same run: sgd_update reads 3 x 135883 x 4 B = 1,630,596 B and writes "
                "2 x 135883 x 4 B = 1,087,064 B." % pre,
    "workload": {"minibatch_per_gpu": 64, "T": 200},
    "kernels": kern,
    "legs": legs,
}
json.dump(doc, open(os.path.join(dst, "pmc_r%s.json" % rnd), "w"), indent=1)
print(json.dumps(kern, indent=1))
