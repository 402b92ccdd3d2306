#!/bin/bash
# Run ON THE GPU BOX from the repo root:  tools/profile_round.sh r03 [config ...]      (default configs: 2 3 5 6 7)
# Per config C (BASELINE.json config number; 2 = the bench default) it leaves under gpurun_out/prof_<tag>/cfgC/:
#   bench.json                 the plain bench line (python3 bench.py --config C ...)
#   trace/ ... kernel_stats    rocprofv3 --kernel-trace --stats of the same command
#   pmc_fetch/, pmc_write/     rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, no tracing)
#   pmc_sq1/, pmc_sq2/         SQ counters (wave cycles, waits, instruction mix, LDS activity / bank conflicts)
#   pmc_flops/                 the hardware's own count of vector flops (SQ_INSTS_VALU_FLOPS_FP32 / _FP64: per lane, x64 per wave)
#   phase_cycles.txt           in-kernel stamps (config 2 only)
# and gpurun_out/prof_<tag>/pmc_summary.json, the list bench.py replays as roofline.traffic.
# The program after `--` is python3 itself (the profiler's preloaded library initialises the GPU: no env/bash hop).
set -o pipefail
tag=${1:-r02}; shift
cfgs=${@:-2 3 5 6 7}
out=$PWD/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
repo=$PWD
for c in $cfgs; do
  o=$out/cfg$c; mkdir -p "$o"
  extra=""; [ "$c" = 5 ] && extra="--batch 8192"
  [ "$c" = 3 ] || [ "$c" = 6 ] || [ "$c" = 7 ] && extra="--batch 4096"
  python3 bench.py --config $c $extra --steps 20 --warmup 3 $([ "$c" = 2 ] || echo "--cpu-sample 0") > "$o/bench.json" 2> "$o/bench.err" || { echo "bench cfg$c failed"; tail -5 "$o/bench.err"; exit 1; }
  echo "cfg$c bench ok: $(cut -c1-160 $o/bench.json)"
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$o/trace" -- python3 "$repo/bench.py" --config $c $extra --steps 20 --warmup 3 --cpu-sample 0 --skip-host-path > "$o/bench_under_rocprof.json" 2> "$o/trace.err") || { echo "trace cfg$c failed"; tail -5 "$o/trace.err"; exit 1; }
  echo "cfg$c trace ok"
  for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" \
              "sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
              "sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_SCA" \
              "flops SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_VALU_FLOPS_FP32_TRANS SQ_INSTS_VALU_FLOPS_FP64_TRANS"; do
    set -- $pass; name=$1; shift
    (cd /tmp && rocprofv3 --pmc $@ --output-format csv -d "$o/pmc_$name" -- python3 "$repo/bench.py" --config $c $extra --steps 3 --warmup 1 --cpu-sample 0 --skip-host-path > /dev/null 2> "$o/pmc_$name.err") || { echo "pmc $name cfg$c failed"; tail -3 "$o/pmc_$name.err"; }
    echo "cfg$c pmc $name done"
  done
  if [ "$c" = 2 ]; then
    python3 tools/phase_cycles.py 4096 2> /dev/null | grep -v amdgpu.ids > "$o/phase_cycles.txt"
  fi
  # in-kernel stamps of whichever family the bench line ran
  hh=$(python3 -c "import json;print(json.load(open('$o/bench.json'))['config']['horizon'])")
  pp=$(python3 -c "import json;print({'dense':1,'stage':2}[json.load(open('$o/bench.json'))['config']['path']])")
  python3 tools/stage_probe.py prof 4096 $hh $pp 2> /dev/null | grep -v amdgpu.ids > "$o/phase_cycles_path.txt"
done
python3 tools/pmc_summary.py "$out" "$tag"
echo "done: $out"
