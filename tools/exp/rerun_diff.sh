#!/bin/bash
# run-to-run reproducibility of one full-size evaluation (and a VAE decode) in separate processes
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do python tests/_fullsize_eval.py /tmp/r$i.npz; done
CCEDIT_SPLIT_CFG=0 CCEDIT_OVERLAP_CONTROLNET=0 python tests/_fullsize_eval.py /tmp/r4.npz
python - <<'PY'
import numpy as np
def rel(a,b): a=a.astype(np.float64); b=b.astype(np.float64); return float(np.sqrt(((a-b)**2).mean())/np.sqrt((b**2).mean()))
r=[np.load(f'/tmp/r{i}.npz') for i in (1,2,3,4)]
for k in ('eps','eps_same','eps_other','frames'):
    print(k, 'runs 1-2 %.2e  1-3 %.2e  1-vs-single-stream %.2e'%(rel(r[0][k], r[1][k]), rel(r[0][k], r[2][k]), rel(r[0][k], r[3][k])))
print('identical halves: two streams %.2e, batched %.2e'%(rel(r[0]['eps_same'][0], r[0]['eps_same'][1]), rel(r[3]['eps_same'][0], r[3]['eps_same'][1])))
print('half 0 beside another clip: %.2e'%rel(r[0]['eps_other'][0], r[0]['eps'][0]))
PY
