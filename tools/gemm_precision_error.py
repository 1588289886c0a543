import sys, os, numpy as np, torch
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,R); sys.path.insert(0,R+'/tests/golden'); sys.path.insert(0,R+'/tests')
import cases
from detweights import det_state_dict, det_wave
import puresound_amd.nnet as PA
name='cfg2_full'; c=cases.CASES[name]
g=dict(np.load(R+'/tests/golden/cfg2_full.npz'))
m=cases.build(PA.NS,name).eval(); m.load_state_dict(det_state_dict(m)); m.to('cuda')
x=det_wave(c['seed'],c['B'],c['L']).to('cuda')
for prec in ('fp32','bf16x3','fp16x2','bf16'):
    m.masker.set_gemm_precision(prec)
    feats,t=m.encoder.encode_padded(x); mask=m.masker.forward_padded(feats,t)
    pre=m.encoder.decode_padded(feats,t,mask,'relu','none').cpu().numpy()
    post=m.inference(x).cpu().numpy()
    r=g['wav_preclamp']
    print(prec, 'pre-clamp max-rel %.3e  l2-rel %.3e   post-clamp max-rel %.3e' % (np.abs(pre-r).max()/np.abs(r).max(), np.linalg.norm(pre-r)/np.linalg.norm(r), np.abs(post-g['wav']).max()/np.abs(g['wav']).max()))

# the same at the benchmark's size (32 utterances: the persistent kernels): deviation from the exact-fp32 MFMA run
import time
x=det_wave(7,32,64000).to('cuda')
m.masker.set_gemm_precision('fp32'); ref=m.inference(x)
for prec in ('bf16x3','fp16x2'):
    m.masker.set_gemm_precision(prec); y=m.inference(x); torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(5): m.inference(x)
    torch.cuda.synchronize()
    print(prec,'32 x 4 s: max |d| vs fp32 run %.3e, l2-rel %.3e, finite %s, %.2f ms/step' % (float((y-ref).abs().max()), float((y-ref).norm()/ref.norm()), bool(torch.isfinite(y).all()), (time.perf_counter()-t0)/5*1e3))
