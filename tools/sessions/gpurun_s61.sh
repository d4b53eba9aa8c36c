# (on the build container first: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o _ab/seulex_lanes_bench.so tools/seulex_lanes_bench.hip)
mkdir -p gpurun_out/s61
timeout 800 python tools/seulex_lanes_probe.py 20000 > gpurun_out/s61/seulex_lanes_probe2.txt 2>&1
