export TMPDIR=/tmp
echo "=== hint on gate 2"; python tools/fused_timing.py gate=2 2>&1 | grep -v "amdgpu.ids\|by XCD\|latest"
echo "=== full, hint on"; python tools/fused_timing.py full 2>&1 | grep -v "amdgpu.ids\|by XCD\|latest"
