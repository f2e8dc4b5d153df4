timeout 200 python tools/bench_gemm_tn_split.py edge,cfg2,cfg3,cfg4,cfg5 check 2>&1 | grep -v amdgpu.ids
timeout 200 python tools/bench_gemm_tn_split.py cfg2,cfg5 stamps 2>&1 | grep -v amdgpu.ids
