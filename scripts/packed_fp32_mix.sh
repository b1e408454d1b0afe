hipcc --offload-arch=gfx950 -O2 -shared -fPIC -DREPRO_SHARED scripts/repro_packed_fp32_hazard.hip -o /tmp/librepro.so || exit 1
NOPESAC_HIPCC_EXTRA="-Xclang -target-feature -Xclang +packed-fp32-ops" python -m nopesac_amd.build --force > /dev/null 2>&1
echo "== library built WITH packed-f32"
timeout 600 python scripts/repro_packed_fp32_mix.py /tmp/librepro.so 2>&1 | grep -v amdgpu
python -m nopesac_amd.build --force > /dev/null 2>&1
