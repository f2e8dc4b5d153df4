# Same-box A/B of one environment switch on the bench workloads:  bash tools/ab_env.sh VAR A_VALUE B_VALUE [cfg[:ragged] ...]
var=$1; a=$2; b=$3; shift 3
cfgs=${@:-cfg2}
for item in $cfgs; do
  cfg=${item%%:*}; extra=""
  if [ "$item" != "$cfg" ]; then extra="--ragged"; fi
  for rep in 1 2 3; do
    for v in $a $b; do
      ms=$(env $var=$v python bench.py --config $cfg $extra --no-extra --no-roofline --no-cpu-baseline --steps 300 --warmup 40 2>/dev/null | python -c "import json,sys; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])")
      echo "$item $var=$v rep$rep $ms ms"
    done
  done
done
