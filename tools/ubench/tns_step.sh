export MMDFN_TUNING_LIB=1
for cfg in cfg2 cfg3 cfg4; do
for sp in 0 1; do
  echo "== $cfg split=$sp"
  MMDFN_TN_SPLIT=$sp timeout 300 python bench.py --config $cfg --steps 200 --warmup 40 --no-extra --no-roofline --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
done; done
