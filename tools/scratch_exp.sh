echo "== diag clock, 1 WG/CU"; RB_TRACE_WGS=256 TOUCAN_RB_WG_PER_CU=1 TOUCAN_HIP_LIB=ims-toucan-prosody-variance_amd/build/variants/libdiagclock.so python tools/microbench_resblock.py --store bf16 --channels 64,32 --taps 7 --acts snake --reps 3
echo "== diag clock, 2 WG/CU"; TOUCAN_HIP_LIB=ims-toucan-prosody-variance_amd/build/variants/libdiagclock.so python tools/microbench_resblock.py --store bf16 --channels 128,64,32 --taps 7 --acts snake --reps 3
echo "== diag clock, 2 WG/CU lrelu-only timing"; python tools/microbench_resblock.py --store bf16 --channels 128,64,32 --taps 7 --reps 5
