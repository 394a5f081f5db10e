#!/bin/bash
# Runs on the MI355X box (through gpurun): collects every artefact profiles/ holds for one version tag.
#   tools/collect_profiles.sh v17     -> gpurun_out/profiles_v17/*
set -e
tag=$1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/profiles_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_bf16_default.json 2> $O/bench_bf16_default.log
python3 $R/bench.py --dtype fp32 --no-cpu-baseline > $O/bench_fp32.json 2>/dev/null
python3 $R/bench.py --vocoder hifigan --no-cpu-baseline > $O/bench_bf16_hifigan.json 2>/dev/null
python3 $R/tools/latency_configs.py > $O/latency_configs.jsonl 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bf16 -o r -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $O/rocprof_bf16.log 2>&1
cp /tmp/p_bf16/*kernel_stats.csv $O/bench_bf16_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_fp32 -o r -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --dtype fp32 > $O/rocprof_fp32.log 2>&1
cp /tmp/p_fp32/*kernel_stats.csv $O/bench_fp32_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_f -o r -- python3 $R/tools/microbench_resblock.py --store bf16 --reps 2 > $O/pmc_fetch.log 2>&1
cp /tmp/p_f/*counter_collection.csv $O/pmc_resblock_FETCH_SIZE.csv
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_w -o r -- python3 $R/tools/microbench_resblock.py --store bf16 --reps 2 > $O/pmc_write.log 2>&1
cp /tmp/p_w/*counter_collection.csv $O/pmc_resblock_WRITE_SIZE.csv
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/p_sq -o r -- python3 $R/tools/microbench_resblock.py --store bf16 --reps 2 > $O/pmc_sq.log 2>&1
cp /tmp/p_sq/*counter_collection.csv $O/pmc_resblock_SQ.csv
python3 $R/tools/pmc_traffic.py $O/pmc_resblock_FETCH_SIZE.csv $O/pmc_resblock_WRITE_SIZE.csv $O/pmc_resblock_traffic.json > $O/pmc_traffic_summary.txt
rm -f $O/*.log
echo done
