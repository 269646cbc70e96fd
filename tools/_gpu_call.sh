bash tools/mff_variants.sh 2>&1 | grep -v amdgpu.ids
