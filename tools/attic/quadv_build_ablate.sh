#!/bin/bash
# builds nextgen-uia_amd/uia_hip/libuia_hip_ablate.so: the shipped library with csrc/gemm_quadv.hip compiled under -DUIA_QUADV_ABLATIONS (the K-loop / epilogue
# ablations of tile cfg 27, selected by UIA_QUADV_ABLATE; results WRONG by construction).  Run where hipcc is (cross-compiles without a GPU); the .so travels with gpurun.
set -e
cd "$(dirname "$0")/../../nextgen-uia_amd/csrc"
make -j8 > /dev/null
mkdir -p /tmp/quadv_ablate
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-result -DUIA_QUADV_ABLATIONS -c gemm_quadv.hip -o /tmp/quadv_ablate/gemm_quadv.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../uia_hip/libuia_hip_ablate.so /tmp/quadv_ablate/gemm_quadv.o $(ls *.o | grep -v '^gemm_quadv.o$' | tr '\n' ' ') -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
ls -la ../uia_hip/libuia_hip_ablate.so
