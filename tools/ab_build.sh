#!/bin/bash
# A/B timing of kernel variants on ONE GPU box (boxes differ by 1-2 %, run-to-run noise on a box is ~0.3 %).
#   tools/ab_build.sh build NAME [extra hipcc flags]   -> build_tmp/NAME.so from the working tree (here, no GPU needed)
#   tools/ab_build.sh run NAME1 NAME2 ...              -> ON THE GPU BOX: three interleaved rounds of bench.py per variant
#                                                         (BENCH_ARGS="--config 5 --batch 8192" for another config)
# build_tmp/ is git-ignored but travels with gpurun, like the in-tree libbmpc.so.  The run leaves the LAST variant
# installed as biped_mpc_py_amd/libbmpc.so on the box only (the box is thrown away after the call).
set -e
cd "$(dirname "$0")/.."
mkdir -p build_tmp
case "$1" in
  build)
    n=$2; shift 2
    (cd biped_mpc_py_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -fPIC -shared -O3 -fno-slp-vectorize \
        -I../../include "$@" bmpc_capi.hip -o ../../build_tmp/$n.so)
    ;;
  run)
    shift
    for rep in 1 2 3; do
      for v in "$@"; do
        cp build_tmp/$v.so biped_mpc_py_amd/libbmpc.so
        python bench.py --steps 30 --warmup 5 --cpu-sample 0 $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-12s' % '$v', '%.4f ms/step kernel %.4f ms iters %.2f max %d' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['mean_iters'], d['config']['max_iters']))"
      done
    done
    ;;
  *) echo "usage: $0 build NAME [flags] | run NAME..."; exit 2;;
esac
