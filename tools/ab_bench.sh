#!/bin/bash
# Developer aid, runs on the GPU box: bench.py's isolated stage times for several builds of the library.
# usage: bash tools/ab_bench.sh lib1.so lib2.so ...   (the in-tree library is measured first)
cp orb_slam2_amd/liborbgpu.so /tmp/orig.so
for so in /tmp/orig.so "$@"; do
  cp "$so" orb_slam2_amd/liborbgpu.so
  timeout -k 10 120 python bench.py --no-latency --no-cpu --steps 10 > gpurun_out/ab.log 2>&1 || { tail -3 gpurun_out/ab.log; exit 1; }
  echo "$so: $(python tools/show_bench.py gpurun_out/ab.log)"
done
cp /tmp/orig.so orb_slam2_amd/liborbgpu.so
