# does the ORIGINAL observation still hold on this box / ROCm?  library with packed-f32 ENABLED vs the shipped build (disabled)
NOPESAC_HIPCC_EXTRA="-Xclang -target-feature -Xclang +packed-fp32-ops" python -m nopesac_amd.build --force > /dev/null 2>&1
echo "== packed-f32 ENABLED build: $(/opt/rocm/lib/llvm/bin/llvm-objdump -d --offloading nopesac_amd/libnopesac_hip.so 2>/dev/null | grep -c v_pk_fma_f32) v_pk_fma_f32"
timeout 600 python scripts/lds_victim.py 16 2>&1 | grep -v amdgpu | grep "launches off"
python -m nopesac_amd.build --force > /dev/null 2>&1
echo "== shipped build (packed-f32 disabled)"
timeout 600 python scripts/lds_victim.py 16 2>&1 | grep -v amdgpu | grep "launches off"
