import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from gritlm_amd import ops
D=128; g=torch.Generator(device="cuda").manual_seed(7)
for (B,S,nq,nkv) in [(1,64,4,4)]:
    qkv=torch.randn((B*S,(nq+2*nkv)*D),generator=g,device="cuda").to(torch.bfloat16)
    bits=ops.mask_pack(torch.ones((B,S),dtype=torch.int64,device="cuda"))
    os.environ["GRIT_ATTN_FWD"]="w64"
    out=torch.zeros((B*S,nq*D),dtype=torch.bfloat16,device="cuda"); lse=torch.zeros((B,nq,S),dtype=torch.float32,device="cuda")
    ops.attn_bidir(qkv,bits,B,S,nq,nkv,D,out=out,lse=lse); torch.cuda.synchronize()
    c=(D**-0.5)*1.4426950408889634
    def exp_l(h):
        s=(qkv[:S,h*D:(h+1)*D].float() @ qkv[:S,(nq+h)*D:(nq+h+1)*D].float().T)*c
        return torch.exp2(s - s.max(1,keepdim=True)[0]).sum(1)
    for h,name in ((0,"l_tot"),(1,"l_run lanes(hi=0)"),(2,"psum(hi=0)"),(3,"m_run")):
        print(name, "rows 0-15:", [round(x,2) for x in lse[0,h,:16].tolist()], " rows 32-39:", [round(x,3) for x in lse[0,h,32:40].tolist()])
    print("expected l head0:", [round(x,3) for x in exp_l(0)[:8].tolist()], "head1:", [round(x,3) for x in exp_l(1)[:8].tolist()], "head2:", [round(x,3) for x in exp_l(2)[:8].tolist()])
