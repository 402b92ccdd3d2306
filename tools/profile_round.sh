#!/bin/bash
# Run ON THE GPU BOX from the repo root:  tools/profile_round.sh r02 [config ...]      (default configs: 2 3 5)
# Per config C (BASELINE.json config number; 2 = the bench default) it leaves under gpurun_out/prof_<tag>/cfgC/:
#   bench.json                 the plain bench line (python3 bench.py --config C ...)
#   trace/ ... kernel_stats    rocprofv3 --kernel-trace --stats of the same command
#   pmc_fetch/, pmc_write/     rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, no tracing)
#   pmc_sq1/, pmc_sq2/         SQ counters (wave cycles, waits, instruction mix, LDS activity / bank conflicts)
#   phase_cycles.txt           in-kernel stamps (config 2 only)
# and gpurun_out/prof_<tag>/pmc_summary.json, the list bench.py replays as roofline.traffic.
# The program after `--` is python3 itself (the profiler's preloaded library initialises the GPU: no env/bash hop).
set -o pipefail
tag=${1:-r02}; shift
cfgs=${@:-2 3 5}
out=$PWD/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
repo=$PWD
for c in $cfgs; do
  o=$out/cfg$c; mkdir -p "$o"
  extra=""; [ "$c" != 2 ] && extra="--batch 8192"
  [ "$c" = 3 ] && extra="--batch 4096"
  python3 bench.py --config $c $extra --steps 20 --warmup 3 --cpu-sample $([ "$c" = 2 ] && echo 256 || echo 0) > "$o/bench.json" 2> "$o/bench.err" || { echo "bench cfg$c failed"; tail -5 "$o/bench.err"; exit 1; }
  echo "cfg$c bench ok: $(cut -c1-160 $o/bench.json)"
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$o/trace" -- python3 "$repo/bench.py" --config $c $extra --steps 20 --warmup 3 --cpu-sample 0 > "$o/bench_under_rocprof.json" 2> "$o/trace.err") || { echo "trace cfg$c failed"; tail -5 "$o/trace.err"; exit 1; }
  echo "cfg$c trace ok"
  for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" \
              "sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
              "sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_SCA"; do
    set -- $pass; name=$1; shift
    (cd /tmp && rocprofv3 --pmc $@ --output-format csv -d "$o/pmc_$name" -- python3 "$repo/bench.py" --config $c $extra --steps 3 --warmup 1 --cpu-sample 0 > /dev/null 2> "$o/pmc_$name.err") || { echo "pmc $name cfg$c failed"; tail -3 "$o/pmc_$name.err"; }
    echo "cfg$c pmc $name done"
  done
  if [ "$c" = 2 ]; then
    python3 tools/phase_cycles.py 4096 2> /dev/null | grep -v amdgpu.ids > "$o/phase_cycles.txt"
  fi
done
python3 - "$out" "$tag" $cfgs <<'PY'
import csv, glob, json, os, sys
out, tag, cfgs = sys.argv[1], sys.argv[2], [int(c) for c in sys.argv[3:]]
H = {2: 10, 3: 16, 4: 10, 5: 20}
def counters(d):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "solve_kernel" in r["Kernel_Name"]:
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
    fe, wr = counters(os.path.join(o, "pmc_fetch")), counters(os.path.join(o, "pmc_write"))
    s = {"config": c, "batch": B, "horizon": H[c], "source": "profiles/%s_cfg%d_pmc_*.csv" % (tag, c)}
    if "FETCH_SIZE" in fe and "WRITE_SIZE" in wr:
        s["fetch_bytes_per_launch"] = 1024.0 * fe["FETCH_SIZE"][0]
        s["write_bytes_per_launch"] = 1024.0 * wr["WRITE_SIZE"][0]
        s["traffic_bytes_per_launch"] = s["fetch_bytes_per_launch"] + s["write_bytes_per_launch"]
        s["traffic_bytes_per_solve"] = s["traffic_bytes_per_launch"] / B
        s["algorithmic_bytes_per_solve"] = line["roofline"]["hbm_algorithmic_bytes_per_solve"]
        s["note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (units KB -> x1024), mean per launch. "
                     "FETCH_SIZE is NOT doubled: the gfx950 x2 correction is calibrated for 16-B/lane coalesced streams, this kernel reads dwords.")
    sq = {}
    for p in ("pmc_sq1", "pmc_sq2"):
        sq.update({k: v[0] for k, v in counters(os.path.join(o, p)).items()})
    s["sq_per_launch"] = sq
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
PY
echo "done: $out"
