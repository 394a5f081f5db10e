#!/bin/bash
# Builds an A/B variant of libtoucan_hip.so: tools/build_variant.sh NAME [extra hipcc flags ...]
# -> ims-toucan-prosody-variance_amd/build/variants/libNAME.so ; run with TOUCAN_HIP_LIB=<that path>.
# (Timings from different gpurun calls differ by up to 10 %: always compare variants inside ONE call.)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
P=ims-toucan-prosody-variance_amd
mkdir -p $P/build/variants/$name
for s in conv1d resblock rowops attention_mfma sequence_ops capi pipeline style wavenet ffn; do
  extra="$*"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra -c $P/csrc/$s.hip -o $P/build/variants/$name/$s.o 2>/dev/null &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/build/variants/lib$name.so $P/build/variants/$name/*.o
echo built $P/build/variants/lib$name.so
