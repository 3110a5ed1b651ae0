import sys, os
sys.argv=[sys.argv[0],"none"]
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
sys.path.insert(0, os.path.join(ROOT,"tools"))
import importlib.util
spec=importlib.util.spec_from_file_location("cb", os.path.join(ROOT,"tools","cgemm_bench.py"))
cb=importlib.util.module_from_spec(spec); spec.loader.exec_module(cb)
import torch
from scnattn._lib import ConvExtra, lib
dev=cb.dev
for (R,Cin,Cout) in [(8192,1024,256),(2048,2048,512),(8192,256,1024),(2048,512,2048),(32768,512,128)]:
    x=torch.randn(R,Cin,device=dev); w=0.1*torch.randn(Cout,Cin,device=dev); y=torch.empty(R,Cout,device=dev)
    dy=torch.randn(R,Cout,device=dev); dx=torch.empty(R,Cin,device=dev)
    part=torch.empty(2,max(Cin,Cout),lib().scnattn_cgemm_stat_ld(R),device=dev)
    row=[]
    for fs in (1,2,3,4):
        ex=ConvExtra(epi=1,stat_partial=part.data_ptr(),force_split=fs)
        exd=ConvExtra(force_split=fs)
        try:
            f=cb.t_us(lambda: cb.cgemm(x,w,False,True,y,R,Cout,Cin,ex))
            d=cb.t_us(lambda: cb.cgemm(dy,w,False,False,dx,R,Cin,Cout,exd))
        except Exception as e:
            f=d=float('nan')
        row.append((f,d))
    print("R=%6d Cin=%5d Cout=%5d | fwd+stats at split 1..4: %s | dgrad: %s" % (R,Cin,Cout," ".join("%6.1f"%r[0] for r in row)," ".join("%6.1f"%r[1] for r in row)), flush=True)

# weight gradients of layer3 / layer2 (K = rows): forced split, two-launch protocol vs in-launch combine
from scnattn import functional as SF
print("wgrad dW[Cout][Cin] = dy^T x: us at forced split (two launches | in-launch combine)")
for (R, Cin, Cout) in [(8192, 1024, 256), (8192, 256, 1024), (32768, 512, 128), (32768, 128, 512), (2048, 2048, 512)]:
    x = torch.randn(R, Cin, device=dev); dy = torch.randn(R, Cout, device=dev); dw = torch.empty(Cout, Cin, device=dev)
    out = []
    for fs in (4, 8, 16, 32):
        row = []
        for comb in (0, 1):
            SF.set_option("cgemm_combine", comb); SF.set_option("cgemm_combine_max", 64)
            try:
                row.append(cb.t_us(lambda: cb.cgemm(dy, x, True, False, dw, Cout, Cin, R, ConvExtra(force_split=fs))))
            except Exception:
                row.append(float("nan"))
        out.append("S=%-2d %5.1f | %5.1f" % (fs, row[0], row[1]))
    print("R=%6d Cin=%5d Cout=%5d | %s" % (R, Cin, Cout, "   ".join(out)), flush=True)
SF.set_option("cgemm_combine", 1); SF.set_option("cgemm_combine_max", 8)
