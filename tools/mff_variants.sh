#!/bin/bash
# Where does the fused Mona forward (one workgroup per image) spend its time?  Builds diagnostic variants of libuia_hip.so that return after
# phase k (results are WRONG in them) and times the kernel at the ViT-B/16 shape against the four unfused launches.
# Run on the GPU box: bash tools/mff_variants.sh
cd $GRAFT_REPO_ROOT/nextgen-uia_amd/csrc
mkdir -p /tmp/mff
OBJS=$(ls *.o | grep -v "^mona_fused.o$" | tr "
" " ")     # every object of the library but the one rebuilt here
for v in ${MFF_VARIANTS:-0 1 2 3 99}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DMF_STOP=$v -c mona_fused.hip -o /tmp/mff/mf_$v.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/mff/lib_$v.so /tmp/mff/mf_$v.o $OBJS -L/opt/rocm/lib -lrccl
  UIA_HIP_LIB=/tmp/mff/lib_$v.so MF_V=$v python3 $GRAFT_REPO_ROOT/tools/time_mona_fused.py
done
