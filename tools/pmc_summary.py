#!/usr/bin/env python3
"""Summary of the rocprofv3 --pmc passes of tools/profile_round.sh: python tools/pmc_summary.py gpurun_out/prof_<tag> <tag>
writes <dir>/pmc_summary.json (what bench.py replays as roofline.traffic) for every config directory found."""
import csv, glob, json, os, sys
out, tag = sys.argv[1], sys.argv[2]
cfgs = sorted(int(os.path.basename(d)[3:]) for d in glob.glob(os.path.join(out, "cfg*")))      # every config ever profiled under this tag
H = {2: 10, 3: 16, 4: 10, 5: 20, 6: 32, 7: 40}
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from biped_mpc_py_amd.synth import kernel_source_hash
def counters(d, B=None, family=None):
    acc = {}
    fs = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for f in fs[-1:]:                            # the newest run only (gpurun_out accumulates every call's files)
        for r in csv.DictReader(open(f)):
            # the kernel family of the line's path only (`--path best` times BOTH families before the timed region)
            if (family or "solve_kernel") in r["Kernel_Name"] or (family is None and "stage_kernel" in r["Kernel_Name"]):
                # whole-batch launches only (the host-pointer path solves a batch in chunks)
                if B is not None and int(r["Grid_Size"]) != B * int(r["Workgroup_Size"]):
                    continue
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}
summ = []
for c in cfgs:
    o = os.path.join(out, "cfg%d" % c)
    try:
        line = json.load(open(os.path.join(o, "bench.json")))
    except Exception:
        continue
    B = line["config"]["batch_per_gpu"]
    fam = "stage_kernel" if line["config"].get("path", "dense") == "stage" else "solve_kernel"
    fe, wr = counters(os.path.join(o, "pmc_fetch"), B, fam), counters(os.path.join(o, "pmc_write"), B, fam)
    s = {"config": c, "batch": B, "horizon": H[c], "source": "profiles/%s_cfg%d_pmc_*.csv" % (tag, c),
         "path": line["config"].get("path", "dense"), "kernel_sha": kernel_source_hash()}
    if "FETCH_SIZE" in fe and "WRITE_SIZE" in wr:
        s["fetch_bytes_per_launch"] = 1024.0 * fe["FETCH_SIZE"][0]
        s["write_bytes_per_launch"] = 1024.0 * wr["WRITE_SIZE"][0]
        s["traffic_bytes_per_launch"] = s["fetch_bytes_per_launch"] + s["write_bytes_per_launch"]
        s["traffic_bytes_per_solve"] = s["traffic_bytes_per_launch"] / B
        s["algorithmic_bytes_per_solve"] = line["roofline"]["hbm_algorithmic_bytes_per_solve"]
        s["note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (units KB -> x1024), mean per launch. "
                     "FETCH_SIZE is NOT doubled: the gfx950 x2 correction is calibrated for 16-B/lane coalesced streams, this kernel reads dwords.")
    sq = {}
    for p in ("pmc_sq1", "pmc_sq2", "pmc_flops"):
        sq.update({k: v[0] for k, v in counters(os.path.join(o, p), B, fam).items()})
    s["sq_per_launch"] = sq
    if "SQ_INSTS_VALU_FLOPS_FP32" in sq:
        # the counters count flops per LANE of a wave instruction (FMA 2, packed FMA 4, ...), whatever the EXEC mask: x 64
        s["valu_flops_per_launch"] = {"f32": 64.0 * (sq["SQ_INSTS_VALU_FLOPS_FP32"] + sq.get("SQ_INSTS_VALU_FLOPS_FP32_TRANS", 0.0)),
                                      "f64": 64.0 * (sq.get("SQ_INSTS_VALU_FLOPS_FP64", 0.0) + sq.get("SQ_INSTS_VALU_FLOPS_FP64_TRANS", 0.0)),
                                      "note": "SQ_INSTS_VALU_FLOPS_FP32/_FP64 (+ _TRANS) x 64 lanes: every vector flop the kernel issues, "
                                              "redundant and masked-lane work included (an upper bound of the useful flops)"}
    summ.append(s)
json.dump(summ, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
for s in summ:
    sq = s["sq_per_launch"]
    print("cfg", s["config"], "traffic/solve", s.get("traffic_bytes_per_solve"), "alg", s.get("algorithmic_bytes_per_solve"))
    if "SQ_WAVE_CYCLES" in sq:
        wc = sq["SQ_WAVE_CYCLES"]
        print("   of wave cycles: wait_any %.2f wait_inst_any %.2f active_inst_any %.2f | valu active %.2f lds active %.2f wait_inst_lds %.2f" % tuple(
            sq.get(k, 0) / wc for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS")))
    if "SQ_INSTS_VALU" in sq:
        print("   per solve: VALU %.0f LDS %.0f SALU %.0f | LDS idx active %.0f bank conflict %.0f (%.1f %%)" % (
            sq["SQ_INSTS_VALU"] / s["batch"], sq.get("SQ_INSTS_LDS", 0) / s["batch"], sq.get("SQ_INSTS_SALU", 0) / s["batch"],
            sq.get("SQ_LDS_IDX_ACTIVE", 0) / s["batch"], sq.get("SQ_LDS_BANK_CONFLICT", 0) / s["batch"],
            100.0 * sq.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, sq.get("SQ_LDS_IDX_ACTIVE", 1))))
