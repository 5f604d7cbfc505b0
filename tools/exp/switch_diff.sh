#!/bin/bash
# which kernel-policy switch moves the full-size prediction how much (rel. RMS against the default fast path)
cd $GRAFT_REPO_ROOT
python tests/_fullsize_eval.py /tmp/fast.npz
python tests/_fullsize_eval.py /tmp/fast2.npz
for sw in CCEDIT_T6 CCEDIT_CONV_HALO CCEDIT_ATTN_SHORT CCEDIT_SPLIT_CFG CCEDIT_OVERLAP_CONTROLNET CCEDIT_KROT CCEDIT_CGROUP CCEDIT_FUSE_GN_STATS CCEDIT_TEMPORAL_ORDER; do
  env $sw=0 python tests/_fullsize_eval.py /tmp/$sw.npz
done
python - <<'PY'
import numpy as np, glob
f=np.load('/tmp/fast.npz')
def rel(a,b): a=a.astype(np.float64); b=b.astype(np.float64); return float(np.sqrt(((a-b)**2).mean())/np.sqrt((b**2).mean()))
for p in sorted(glob.glob('/tmp/*.npz')):
    g=np.load(p); print(p, 'eps %.5f frames %.5f'%(rel(g['eps'],f['eps']), rel(g['frames'],f['frames'])))
print('eps rms', float(np.sqrt((f['eps'].astype(np.float64)**2).mean())))
PY
