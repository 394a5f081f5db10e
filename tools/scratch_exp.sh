V=ims-toucan-prosody-variance_amd/build/variants
echo "== old, diag"; TOUCAN_HIP_LIB=$V/libolddiag.so python tools/microbench_resblock.py --store bf16 --channels 64 --taps 11 --acts snake,lrelu --reps 3
echo "== new, diag"; TOUCAN_HIP_LIB=$V/libdiagclock.so python tools/microbench_resblock.py --store bf16 --channels 64 --taps 11 --acts snake,lrelu --reps 3
