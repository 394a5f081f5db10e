#!/bin/bash
# Runs on the MI355X box (through gpurun): collects every artefact profiles/ holds for one tag.
#   tools/collect_profiles.sh r02     -> gpurun_out/profiles_r02/*   (copy what is to be judged into profiles/, prefixed with the tag)
set -e
tag=$1
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/profiles_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
echo "[collect] bench lines"; date
python3 $R/bench.py > $O/bench_bf16_default.json 2> $O/bench_bf16_default.log
python3 $R/bench.py --dtype fp16 --pitch-scale 1.3 --energy-scale 0.7 --no-cpu-baseline > $O/bench_fp16_configs4.json 2>/dev/null
python3 $R/bench.py --dtype fp32 --no-cpu-baseline > $O/bench_fp32.json 2>/dev/null
python3 $R/bench.py --vocoder hifigan --no-cpu-baseline > $O/bench_bf16_hifigan.json 2>/dev/null
python3 $R/bench.py --sequencer python --no-cpu-baseline > $O/bench_bf16_python_sequencer.json 2>/dev/null
python3 $R/bench.py --no-overlap --no-cpu-baseline > $O/bench_bf16_one_stream.json 2>/dev/null
python3 $R/bench.py --dtype mixed --no-cpu-baseline > $O/bench_mixed_f32_f16.json 2>/dev/null
python3 $R/bench.py --dtype mixed3 --no-cpu-baseline > $O/bench_mixed3_f32x3_f16.json 2>/dev/null
python3 $R/bench.py --dtype fp32x3 --no-cpu-baseline > $O/bench_fp32x3.json 2>/dev/null
python3 $R/tools/latency_configs.py > $O/latency_configs.jsonl 2>/dev/null
python3 $R/tools/stage_times.py > $O/stage_times_bf16.txt 2>/dev/null
python3 $R/tools/stage_times.py --batch 1 --precision f32 > $O/stage_times_b1_f32.txt 2>/dev/null
python3 $R/tools/microbench_small.py > $O/microbench_small_f32.txt 2>/dev/null
python3 $R/tools/conv_shapes.py > $O/conv_shapes_bf16.txt 2>/dev/null
python3 $R/tools/microbench_ffn.py > $O/microbench_ffn_bf16.txt 2>/dev/null
echo "[collect] kernel traces"; date
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bf16 -o r -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-verify > $O/rocprof_bf16.log 2>&1
cp /tmp/p_bf16/*kernel_stats.csv $O/bench_bf16_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_b1 -o r -- python3 $R/tools/latency_configs.py --reps 5 > $O/rocprof_b1.log 2>&1
cp /tmp/p_b1/*kernel_stats.csv $O/latency_configs_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_fp16 -o r -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-verify --dtype fp16 --pitch-scale 1.3 --energy-scale 0.7 > $O/rocprof_fp16.log 2>&1
cp /tmp/p_fp16/*kernel_stats.csv $O/bench_fp16_kernel_stats.csv
echo "[collect] counter passes (separate runs, --kernel-trace only)"; date
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_f -o r -- python3 $R/tools/microbench_resblock.py --store bf16 --reps 2 > $O/pmc_fetch.log 2>&1
cp /tmp/p_f/*counter_collection.csv $O/pmc_resblock_FETCH_SIZE.csv
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_w -o r -- python3 $R/tools/microbench_resblock.py --store bf16 --reps 2 > $O/pmc_write.log 2>&1
cp /tmp/p_w/*counter_collection.csv $O/pmc_resblock_WRITE_SIZE.csv
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/p_sq -o r -- python3 $R/tools/microbench_resblock.py --store bf16 --reps 2 > $O/pmc_sq.log 2>&1
cp /tmp/p_sq/*counter_collection.csv $O/pmc_resblock_SQ.csv
echo "[collect] second SQ pass (co-execution, LDS)"; date
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS --kernel-trace --output-format csv -d /tmp/p_sq2 -o r -- python3 $R/tools/microbench_resblock.py --store bf16 --reps 2 > $O/pmc_sq2.log 2>&1
cp /tmp/p_sq2/*counter_collection.csv $O/pmc_resblock_SQ2.csv
python3 $R/tools/pmc_dump.py $O/pmc_resblock_SQ2.csv resblock > $O/pmc_sq2_dump.txt
python3 $R/tools/pmc_traffic.py $O/pmc_resblock_FETCH_SIZE.csv $O/pmc_resblock_WRITE_SIZE.csv $O/pmc_resblock_traffic.json > $O/pmc_traffic_summary.txt
python3 $R/tools/pmc_sq_summary.py $O/pmc_resblock_SQ.csv $O/pmc_resblock_SQ_summary.json > $O/pmc_sq_summary.txt
python3 $R/tools/microbench_resblock.py --store bf16 --reps 5 > $O/microbench_resblock_bf16.txt 2>/dev/null
rm -f $O/pmc_fetch.log $O/pmc_write.log $O/pmc_sq.log $O/pmc_sq2.log $O/rocprof_bf16.log $O/rocprof_fp16.log $O/rocprof_b1.log
echo "[collect] done"; date
