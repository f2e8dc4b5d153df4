#!/bin/bash
# builds tools/k6_planes/bin/libk6planes.so (gfx950); the experiment tools load it with ctypes
set -e
cd "$(dirname "$0")"
mkdir -p bin
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I ../../include -o bin/libk6planes.so propagate_planes.hip propagate_planes2.hip
echo built bin/libk6planes.so
