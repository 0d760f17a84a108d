#!/bin/bash
# On the GPU box: SQ counters of the attention kernel (one --pmc pass per argument, each a quoted counter list).
# usage: tools/pmc_attn.sh TAG "SQ_WAVE_CYCLES SQ_WAIT_ANY ..." ["..."]    env (e.g. DIFFAB_HIP_LIB, DIFFAB_ATTN_FLASH) is inherited
R=$GRAFT_REPO_ROOT; tag=$1; shift
O=$R/gpurun_out/r2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0; dirs=""
for pass in "$@"; do
  i=$((i+1)); rm -rf /tmp/pmc_$tag$i
  timeout -k 10 240 rocprofv3 --pmc $pass --output-format csv -d /tmp/pmc_$tag$i -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs $BENCH_ARGS > $O/pmc_$tag$i.log 2>&1 || { echo "pass $i failed"; tail -3 $O/pmc_$tag$i.log; exit 1; }
  dirs="$dirs /tmp/pmc_$tag$i"
done
python3 $R/tools/pmc_summarise.py $O/pmc_$tag.json $dirs | grep -E "${PMC_GREP:-attn}"
