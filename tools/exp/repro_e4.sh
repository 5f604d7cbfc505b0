export TRIALS=2 ROUNDS=150
LOADS=t1,t2,t3,t4,t5,t6 python tools/exp/repro_e4.py 2>&1 | grep -v amdgpu
for c in silu gn torch; do CONS=$c LOADS=t1 python tools/exp/repro_e4.py 2>&1 | grep -v amdgpu; done
CCEDIT_HIP_LIB=$PWD/build_var/libccedit_ln_noslp.so LOADS=t1,unet TRIALS=3 python tools/exp/repro_e4.py 2>&1 | grep -v amdgpu
