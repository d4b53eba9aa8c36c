# (on the build container first: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o _ab/gbs_lanes_bench.so tools/gbs_lanes_bench.hip)
mkdir -p gpurun_out/s62
timeout 800 python tools/gbs_lanes_probe.py 20000 > gpurun_out/s62/gbs_lanes_probe.txt 2>&1
