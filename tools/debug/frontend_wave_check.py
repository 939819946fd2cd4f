import sys, torch
sys.path.insert(0,'/root/repo')
from transformer4sed_amd import synth
from transformer4sed_amd.frontend import PasstFeatureExtractor
from transformer4sed_amd.ops import call
dev=torch.device("cuda")
ext=PasstFeatureExtractor(fmin_aug_range=10,fmax_aug_range=2000).to(dev).eval()
for B,L in ((3,320000),(2,319999),(1,160001),(2,33000)):
    wav=torch.from_numpy(synth.synth_wav(B,seed=5))[:, :L].contiguous().to(dev)
    T=1+(L-1)//320
    melw,rng=ext._bank(ext.fmin,ext.fmax,dev)
    outs=[]
    for flag in (1,3,0,2):
        out=torch.empty(B,128,T,device=dev); tmp=torch.empty(B*32,dtype=torch.int32,device=dev)
        call("sed_logmel_fwd",wav,out,tmp,ext.window,ext.twiddle,melw,rng,B,L,T,flag)
        outs.append(out)
    print(B,L,T,"log: max|new-old|",float((outs[0]-outs[1]).abs().max()),"raw rel", float(((outs[2]-outs[3]).abs()/(outs[3].abs()+1e-6)).max()), "finite", bool(torch.isfinite(outs[0]).all()))
