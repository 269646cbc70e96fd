#!/bin/bash
# ln_lora_down at one workgroup more per CU (min-waves launch bound, a few spilled registers) against the default build: ViT-L/14 + LoRA bench, same box.
cd $GRAFT_REPO_ROOT/nextgen-uia_amd/csrc
mkdir -p /tmp/lnld
OBJS=$(ls *.o | grep -v "^lora_rank.o$" | tr "\n" " ")
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DLNLD_MIN_WAVES=3 -c lora_rank.hip -o /tmp/lnld/lora_rank.o 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/lnld/lib3.so /tmp/lnld/lora_rank.o $OBJS -L/opt/rocm/lib -lrccl
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  echo -n "default: "; bash tools/bench_ms.sh --config vitl_lora
  echo -n "3 waves: "; UIA_HIP_LIB=/tmp/lnld/lib3.so bash tools/bench_ms.sh --config vitl_lora
done
