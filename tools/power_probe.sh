#!/bin/bash
# Board power and shader clock while one kernel class runs back to back (run on the MI355X box).
#   bash tools/power_probe.sh "<microbench args>" [label]
args="$1"; label="${2:-probe}"
python tools/microbench_resblock.py $args --reps ${REPS:-1500} > /tmp/pp_$label.txt 2>&1 &
pid=$!
sleep ${WARM:-12}
for i in 1 2 3 4 5; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power\|sclk" | tr '\n' ' '; echo
  sleep 1
done
wait $pid
grep -v amdgpu /tmp/pp_$label.txt | tail -2
