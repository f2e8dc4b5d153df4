cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/tr; rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o p -- python /root/repo/bench.py --no-extra --no-roofline --no-cpu-baseline --steps 20 --warmup 5 > /tmp/tr.log 2>&1
cd /root/repo; f=$(ls /tmp/tr/*/p_kernel_trace.csv /tmp/tr/p_kernel_trace.csv 2>/dev/null | head -1); python tools/step_timeline.py $f
