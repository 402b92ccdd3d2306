#!/bin/bash
# Run ON THE GPU BOX from the repo root:  tools/refresh_profiles.sh r01
# Produces under gpurun_out/prof_<tag>/ everything profiles/README.md lists: the plain bench line, the
# rocprofv3 kernel-trace statistics of the same command, the two PMC passes (separate runs, no
# tracing), the in-kernel phase stamps, and the PMC summary bench.py reports as roofline.traffic.
set -o pipefail
tag=${1:-r01}
out=$PWD/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
repo=$PWD
python3 bench.py --steps 20 --warmup 3 > "$out/bench.json" 2> "$out/bench.err" || { echo "bench failed"; tail -5 "$out/bench.err"; exit 1; }
echo "bench ok"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$repo/bench.py" --steps 20 --warmup 3 --cpu-sample 0 > "$out/bench_under_rocprof.json" 2> "$out/trace.err") || { echo "trace failed"; tail -5 "$out/trace.err"; exit 1; }
echo "trace ok"
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 "$repo/bench.py" --steps 3 --warmup 1 --cpu-sample 0 > /dev/null 2> "$out/pmc_fetch.err") || { echo "pmc fetch failed"; tail -5 "$out/pmc_fetch.err"; exit 1; }
echo "pmc fetch ok"
(cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 "$repo/bench.py" --steps 3 --warmup 1 --cpu-sample 0 > /dev/null 2> "$out/pmc_write.err") || { echo "pmc write failed"; tail -5 "$out/pmc_write.err"; exit 1; }
echo "pmc write ok"
python3 tools/phase_cycles.py 256 2> /dev/null | grep -v amdgpu.ids > "$out/phase_cycles_b256.txt"
python3 tools/phase_cycles.py 4096 2> /dev/null | grep -v amdgpu.ids > "$out/phase_cycles_b4096.txt"
python3 - "$out" "$tag" <<'PY'
import csv, glob, json, os, sys
out, tag = sys.argv[1], sys.argv[2]
def counter(d, name):
    vals = []
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and "solve_kernel<10" in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    return vals
fe, wr = counter("pmc_fetch", "FETCH_SIZE"), counter("pmc_write", "WRITE_SIZE")
B = 4096
s = {"batch": B, "launches": [len(fe), len(wr)],
     "fetch_bytes_per_launch": 1024.0 * sum(fe) / max(1, len(fe)),
     "write_bytes_per_launch": 1024.0 * sum(wr) / max(1, len(wr))}
s["traffic_bytes_per_launch"] = s["fetch_bytes_per_launch"] + s["write_bytes_per_launch"]
s["traffic_bytes_per_solve"] = s["traffic_bytes_per_launch"] / B
s["algorithmic_bytes_per_solve"] = 4 * (12 + 6 + 1) + 2 * 10 + 4 * 25 * 10 + 20
s["note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (units KB -> x1024). FETCH_SIZE is "
             "NOT doubled: the gfx950 x2 correction is calibrated for 16-B/lane coalesced streams, this kernel reads dwords.")
json.dump(s, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(s))
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print(open(f).read().splitlines()[1][:40], open(f).read().splitlines()[1].split('",')[1:])
PY
echo "done: $out"
