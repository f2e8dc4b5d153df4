import sys, torch
sys.path.insert(0, "/root/repo")
from mm_dfn_amd import _hip
lib=_hip.lib(); P=_hip.ptr; st=_hip.stream
H=100
for R in (98304, 24576):
    q,h,c=(torch.randn(R,H,device="cuda") for _ in range(3))
    Wih,Whh=torch.randn(4*H,H,device="cuda")*0.2, torch.randn(4*H,H,device="cuda")*0.2
    b1,b2=torch.randn(4*H,device="cuda"),torch.randn(4*H,device="cuda")
    g,ho,co=torch.empty(R,4*H,device="cuda"),torch.empty(R,H,device="cuda"),torch.empty(R,H,device="cuda")
    planes=torch.empty(int(lib.mmdfn_lstm_gate_planes_workspace(H)),device="cuda")
    lib.mmdfn_lstm_gate_cut_weights(P(Wih),P(Whh),P(planes),H,st())
    for name,pl in (("own cut",None),("pre-cut",planes),("own cut",None),("pre-cut",planes)):
        def f(): lib.mmdfn_lstm_gate_fwd_pre(P(q),P(h),P(c),P(Wih),P(Whh),P(b1),P(b2),P(g),P(ho),P(co),R,H,H,P(pl),st())
        for _ in range(5): f()
        torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); e1.synchronize()
        print(R,name,"%.1f us"%(e0.elapsed_time(e1)/20*1e3))
