import os, sys, json, torch
sys.path.insert(0, os.getcwd())
from vista_amd import ops
n=int(sys.argv[1]) if len(sys.argv)>1 else 50
out={}
for C,S,heads in ((320,9216,5),(640,2304,10),(1280,576,20)):
    g=torch.Generator(device="cuda").manual_seed(0)
    qkv=torch.randn(n*S,3*C,device="cuda",generator=g).to(torch.bfloat16)
    pre=os.environ.get('ATTN_LOG2','0')=='1'
    if pre: qkv[:,:C]*=0.18033688
    fn=lambda: ops.attn_spatial(qkv[:,:C],qkv[:,C:2*C],qkv[:,2*C:],n,heads,S,v_rows=True,q_log2=pre)
    fn();fn();torch.cuda.synchronize()
    best=1e9
    for _ in range(4):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record();torch.cuda.synchronize()
        best=min(best,e0.elapsed_time(e1)/5)
    out[f"S={S}"]=round(best,4)
    out[f"TFLOPs_S={S}"]=round(4.0*n*heads*S*S*64/best/1e9)
print(json.dumps(out))
