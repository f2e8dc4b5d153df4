# step time with the plain / valid-length GRU launches (and the all-padding table):  bash tools/ab_gru_truncate.sh
for cfgargs in "--config cfg3 --ragged" "--config cfg4" "--config cfg4 --ragged" "--config cfg2"; do
  for mode in "0 0 0" "auto 0 0" "auto 0 1" "auto 1 1"; do
    set -- $mode
    MMDFN_GRU_TRUNCATE=$1 MMDFN_GRU_TABLE=$2 MMDFN_GRU_L1_SEG=$3 python bench.py $cfgargs --no-extra --no-roofline --no-cpu-baseline --steps 200 --warmup 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('RESULT', '$cfgargs', 'truncate=$1 table=$2 l1seg=$3', round(d['ms_per_step'],4), round(d['value']))"
  done
done
