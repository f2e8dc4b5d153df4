timeout 200 python tools/bench_gemm_tn_split.py edge,cfg2,cfg3,cfg4,cfg5 check 2>&1 | grep -v amdgpu.ids
for wgs in 256 384 768 1024; do echo "WGS=$wgs"; MMDFN_TNS_WGS=$wgs timeout 120 python tools/bench_gemm_tn_split.py cfg2,cfg3,cfg4,cfg5 2>&1 | grep -v amdgpu.ids | sed 's/split=0[^s]*TFLOP.s//'; done
for abl in 1 2 4 7 8 15; do echo "ABL=$abl"; MMDFN_TNS_ABL=$abl timeout 120 python tools/bench_gemm_tn_split.py cfg2,cfg5 2>&1 | grep -v amdgpu.ids | sed 's/split=0[^s]*TFLOP.s//'; done
