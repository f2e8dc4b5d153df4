#!/bin/bash
# builds tools/proj_stream/bin/libprojstream.so (gfx950): the weight-streaming projection experiment of round 5
set -e
cd "$(dirname "$0")"
mkdir -p bin
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o bin/libprojstream.so proj_stream.hip
echo built bin/libprojstream.so
