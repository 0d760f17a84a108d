#!/bin/bash
# On the GPU box: the round's measurement artefacts -> gpurun_out/round/.  usage: tools/profile_round.sh [tag]
# 1. bench.py (default flags: 100 steps + CPU baseline)         -> bench_n1.json
# 2. rocprofv3 --kernel-trace --stats of a short bench run       -> kernel_stats.csv
# 3. rocprofv3 --pmc passes (separate runs, short bench)         -> pmc_counters_per_launch.json
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/round
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
echo "[1] bench"; timeout -k 10 600 python3 $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err || exit 1
tail -c 600 $O/bench_n1.json; echo
echo "[2] kernel stats"; rm -rf /tmp/ks
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o run -- python3 $R/bench.py --steps 20 --warmup 3 --repeats 1 --no-cpu-baseline --no-other-configs > $O/ks.log 2>&1 || exit 1
cp /tmp/ks/run_kernel_stats.csv $O/kernel_stats.csv
echo "[2b] kernel stats, per-layer launches (--multi-launch)"; rm -rf /tmp/ksm
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ksm -o run -- python3 $R/bench.py --multi-launch --steps 20 --warmup 3 --repeats 1 --no-cpu-baseline --no-other-configs > $O/ksm.log 2>&1 || exit 1
cp /tmp/ksm/run_kernel_stats.csv $O/kernel_stats_multi_launch.csv
i=0
for pass in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1)); echo "[3.$i] pmc $pass"; rm -rf /tmp/pmc$i
  timeout -k 10 240 rocprofv3 --pmc $pass --output-format csv -d /tmp/pmc$i -o run -- python3 $R/bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-other-configs > $O/pmc$i.log 2>&1 || { echo "pass $i failed"; tail -3 $O/pmc$i.log; exit 1; }
done
python3 $R/tools/pmc_summarise.py $O/pmc_counters_per_launch.json /tmp/pmc1 /tmp/pmc2 /tmp/pmc3 /tmp/pmc4
# 4. the other legs: training step (config 4 per-GPU share), K = 256 sampling (config 5 shape), encode_context
echo "[4] training kernel stats"; rm -rf /tmp/kst
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o run -- python3 $R/bench.py --train --steps 10 --warmup 2 --no-cpu-baseline > $O/kst.log 2>&1 || exit 1
cp /tmp/kst/run_kernel_stats.csv $O/train_kernel_stats.csv; grep '^{' $O/kst.log > $O/train_bench.json
echo "[5] K=256 kernel stats"; rm -rf /tmp/ks256
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks256 -o run -- python3 $R/bench.py --k 256 --batch 128 --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-other-configs > $O/ks256.log 2>&1 || exit 1
cp /tmp/ks256/run_kernel_stats.csv $O/k256_kernel_stats.csv; grep '^{' $O/ks256.log > $O/k256_bench.json
echo "[6] encode_context"; rm -rf /tmp/ksec
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ksec -o run -- python3 $R/tools/encode_context_bench.py 128 128 --backward > $O/encode_context.log 2>&1 || exit 1
cp /tmp/ksec/run_kernel_stats.csv $O/encode_context_kernel_stats.csv; cat $O/encode_context.log | grep -v rocprof | tail -4
echo done
