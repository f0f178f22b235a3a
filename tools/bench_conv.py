import sys, torch, math, time
import os; R_=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R_); sys.path.insert(0,os.path.join(R_,'ood-gan-inversion_amd'))
from oodgan import ops, synth
dev=torch.device('cuda:0')
shapes=[(8,512,512,64),(8,256,256,128),(8,128,128,256),(8,64,64,512),(8,32,32,1024)]
modes=[('S1',ops.CONV_S1),('T2',ops.CONV_T2),('S2',ops.CONV_S2)]
precs=sys.argv[1].split(',') if len(sys.argv)>1 else ['f16s','f32']
which=sys.argv[2].split(',') if len(sys.argv)>2 else ['S1','T2','S2']
for prec in precs:
  for (B,K,M,H) in shapes:
    w=torch.randn(M,K,3,3,device=dev)/math.sqrt(K*9)
    s=torch.randn(B,K,device=dev)*0.3+1; d=torch.randn(B,M,device=dev)*0.3+1
    for name,mode in modes:
        if name not in which: continue
        if name=='S2':
            Hin=2*H+1; P2=(Hin+3)//4*4; x=torch.randn(B,K,Hin,P2,device=dev); kw=dict(in_hw=(Hin,Hin),in_pitch=P2); flops=2*B*K*M*9*H*H
        elif name=='T2':
            if H>512: continue
            x=torch.randn(B,K,H,H,device=dev); kw={}; flops=2*B*K*M*9*H*H
        else:
            x=torch.randn(B,K,H,H,device=dev); kw={}; flops=2*B*K*M*9*H*H
        wpk=ops.pack_conv3x3(w,1.0,precision=prec)
        y=None
        for it in range(2): y=ops.conv3x3(x,wpk,M,mode,in_scale=s,out_scale=d,**kw)
        torch.cuda.synchronize()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        n=5; e0.record()
        for it in range(n): y=ops.conv3x3(x,wpk,M,mode,in_scale=s,out_scale=d,**kw)
        e1.record(); torch.cuda.synchronize()
        ms=e0.elapsed_time(e1)/n
        print(f'{prec:5s} {name} B{B} K{K} M{M} H{H}: {ms*1e3:8.1f} us  {flops/ms/1e9:7.1f} TF/s', flush=True)
        del x,y
# S-form input variant
if 'SF' in which:
  for (B,K,M,H) in shapes:
    w=torch.randn(M,K,3,3,device=dev)/math.sqrt(K*9)
    s=torch.randn(B,K,device=dev)*0.3+1; d=torch.randn(B,M,device=dev)*0.3+1
    x=torch.randn(B,K,H,H,device=dev)
    xs=ops.to_sform(x,s)
    wpk=ops.pack_conv3x3(w,1.0,precision='f16s')
    ys=ops.SForm(B,M,H,H,dev)
    for variant in ('F-out','F+S-out'):
        kw=dict(out_scale=d) if variant=='F-out' else dict(out_scale=d, ys=ys, ys_scale=d)
        for it in range(2): y=ops.conv3x3(xs,wpk,M,ops.CONV_S1,**kw)
        torch.cuda.synchronize()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        n=5; e0.record()
        for it in range(n): y=ops.conv3x3(xs,wpk,M,ops.CONV_S1,**kw)
        e1.record(); torch.cuda.synchronize()
        ms=e0.elapsed_time(e1)/n; flops=2*B*K*M*9*H*H
        print(f'sform {variant:8s} B{B} K{K} M{M} H{H}: {ms*1e3:8.1f} us  {flops/ms/1e9:7.1f} TF/s', flush=True)
    if H<=512:
        for it in range(2): z=ops.conv3x3(xs,wpk,M,ops.CONV_T2,out_scale=d)
        torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
        for it in range(5): z=ops.conv3x3(xs,wpk,M,ops.CONV_T2,out_scale=d)
        e1.record(); torch.cuda.synchronize(); ms=e0.elapsed_time(e1)/5
        print(f'sform T2       B{B} K{K} M{M} H{H}: {ms*1e3:8.1f} us  {2*B*K*M*9*H*H/ms/1e9:7.1f} TF/s', flush=True)
        del z
    del x,xs,ys,y
if 'SF2' in which:
  for (B,K,M,H) in shapes:
    w=torch.randn(K,M,3,3,device=dev)/math.sqrt(K*9)   # (Co=K_in_bwd..): weight (Co,Ci): bwd K=Co, M=Ci
    d=torch.randn(B,K,device=dev)*0.3+1; s=torch.randn(B,M,device=dev)*0.3+1
    Hin=2*H+1; P2=(Hin+3)//4*4
    g2=torch.randn(B,K,Hin,P2,device=dev)
    wpk=ops.pack_conv3x3(w,1.0,transpose=True,flip=False,precision='f16s')
    gp=ops.to_sform_phases(g2,H,H,d,in_pitch=P2)
    for it in range(2): y=ops.conv3x3(gp,wpk,M,ops.CONV_S2,out_scale=s)
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
    for it in range(5): y=ops.conv3x3(gp,wpk,M,ops.CONV_S2,out_scale=s)
    e1.record(); torch.cuda.synchronize(); ms=e0.elapsed_time(e1)/5
    print(f'sform S2 B{B} K{K} M{M} H{H}: {ms*1e3:8.1f} us  {2*B*K*M*9*H*H/ms/1e9:7.1f} TF/s', flush=True)
    e0.record()
    for it in range(5): gp=ops.to_sform_phases(g2,H,H,d,out=gp,in_pitch=P2)
    e1.record(); torch.cuda.synchronize(); print(f'   to_sform_phases {e0.elapsed_time(e1)/5*1e3:8.1f} us')
    del g2,gp,y
