# Where the time of the fused K6 + K7 forward launch (csrc/gcn_small.hip) goes: MMDFN_PROP_LAYER_STOP=k of the tuning build returns
# early (timing only): 9 launch + decode, 1 requests + LDS fill, 2 + first product, 0 everything.  Per-kernel averages inside the cfg2 step.
for k in 9 1 2 0; do
  echo "stop=$k"; MMDFN_TUNING_LIB=1 MMDFN_PROP_LAYER_STOP=$k bash tools/prof_kernels.sh "prop_layer|propagate_v2|gcnii_layer_fwd" 2>&1 | grep -v "^$"
done
