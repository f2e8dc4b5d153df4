#!/bin/bash
# builds and runs the streaming-read microbenchmark on the GPU box:  tools/bw/run.sh [MB per launch]
cd "$(dirname "$0")" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/read_bw read_bw.hip && /tmp/read_bw ${1:-800}
