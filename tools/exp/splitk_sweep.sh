for s in 2 3 4; do echo "== CCEDIT_G8_SPLIT=$s"; CCEDIT_G8_SPLIT=$s timeout 300 python tools/exp/splitk_perf.py 2>&1 | grep -v amdgpu | head -4; done
