#!/bin/bash
# run on the GPU box: for each lib variant run the bench (no cpu baseline), print value + kernel_ms
for v in "$@"; do
  cp build_tmp/lib_$v.so biped_mpc_py_amd/libbmpc.so
  timeout -k 10 120 python bench.py --cpu-sample 0 > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { echo "$v FAILED"; tail -3 gpurun_out/ab_$v.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_$v.json"))
print("$v", round(d["value"]), "solves/s  kernel_ms", round(d["roofline"]["kernel_ms"],4), "iters", d["config"]["mean_iters"], "nf", d["config"]["mean_factorisations"], "nc", d["config"]["not_converged"])
PY
done
