import os, sys, torch
sys.path.insert(0, '/root/repo')
from nb_asr_amd import hip
DEV='cuda:0'
for cin,cout,tin,s in ((600,768,1024,1),(800,1024,1024,2)):
    b=64
    x=torch.randn(b,cin,tin,device=DEV); w=torch.randn(cout,cin,8,device=DEV)*0.02; bias=torch.randn(cout,device=DEV)
    y=torch.empty(b,cout,hip.round_up4((tin+s-1)//s),device=DEV); packed=hip.pack_dense_weights(w, s)
    os.environ.pop('NBASR_PROF_DUMP', None)
    for _ in range(3): hip.dense_conv1d_fused_packed(x,tin,packed,cout,8,bias,(),y,s)
    torch.cuda.synchronize()
    os.environ['NBASR_PROF_DUMP']='1'
    print('stride', s)
    hip.dense_conv1d_fused_packed(x,tin,packed,cout,8,bias,(),y,s)
    torch.cuda.synchronize()
