#!/bin/bash
# builds tools/k6_pc/bin/libk6pc.so (gfx950): the producer / consumer K6 experiment of round 4, outside the product library
set -e
cd "$(dirname "$0")"
mkdir -p bin
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared ${K6PC_DEFINES} -I ../../include -o bin/libk6pc.so propagate_pc.hip
echo built bin/libk6pc.so
