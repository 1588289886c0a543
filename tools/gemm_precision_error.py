import sys, os, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests/golden'); sys.path.insert(0,'/root/repo/tests')
import cases
from detweights import det_state_dict, det_wave
import puresound_amd.nnet as PA
name='cfg2_full'; c=cases.CASES[name]
g=dict(np.load('/root/repo/tests/golden/cfg2_full.npz'))
m=cases.build(PA.NS,name).eval(); m.load_state_dict(det_state_dict(m)); m.to('cuda')
x=det_wave(c['seed'],c['B'],c['L']).to('cuda')
for prec in ('fp32','bf16x3','bf16'):
    m.masker.set_gemm_precision(prec)
    feats,t=m.encoder.encode_padded(x); mask=m.masker.forward_padded(feats,t)
    pre=m.encoder.decode_padded(feats,t,mask,'relu','none').cpu().numpy()
    post=m.inference(x).cpu().numpy()
    r=g['wav_preclamp']
    print(prec, 'pre-clamp max-rel %.3e  l2-rel %.3e   post-clamp max-rel %.3e' % (np.abs(pre-r).max()/np.abs(r).max(), np.linalg.norm(pre-r)/np.linalg.norm(r), np.abs(post-g['wav']).max()/np.abs(g['wav']).max()))
