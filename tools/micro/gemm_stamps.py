import sys,os,time,torch,ctypes as C
sys.path.insert(0,os.getcwd())
from sdvar_amd import engine as E
lib=E.load_library(); dev=torch.device("cuda:0"); st=C.c_void_p(torch.cuda.current_stream().cuda_stream)
M,N,K=4096,4096,1024
X=torch.randn(M,K,device=dev); W=torch.randn(N,K,device=dev)*0.02; out=torch.empty(M,N,device=dev)
Xp=torch.empty(3,K//32,M,32,dtype=torch.int16,device=dev); Wp=torch.empty(3,K//32,N,32,dtype=torch.int16,device=dev)
P=lambda t:C.c_void_p(t.data_ptr())
E._check(lib.sdvar_op_split_planes(P(X),P(Xp),M,K,M*K,st)); E._check(lib.sdvar_op_split_planes(P(W),P(Wp),N,K,N*K,st))
def run(): E._check(lib.sdvar_op_gemm_bf16x3(P(Xp),M*K,P(Wp),N*K,None,P(out),N,None,0,M,N,K,0,None,0,None,1,0,st))
for _ in range(3): run()
stamps=torch.zeros(1024*4,dtype=torch.int64,device=dev)
E._check(lib.sdvar_debug_set_gemm_stamps(P(stamps)))
run(); torch.cuda.synchronize()
E._check(lib.sdvar_debug_set_gemm_stamps(None))
s=stamps.cpu().view(1024,4).double()
t0=s[:,0].min()
import numpy as np
print("s_memtime ticks (100 MHz?) ; per-WG medians: prologue+tile0 %.0f, loop(31 tiles) %.0f, epilogue %.0f, total %.0f" % ((s[:,1]-s[:,0]).median(), (s[:,2]-s[:,1]).median(), (s[:,3]-s[:,2]).median(), (s[:,3]-s[:,0]).median()))
print("kernel span", (s[:,3].max()-t0).item(), "start spread", (s[:,0].max()-t0).item())
order=torch.argsort(s[:,0]); 
print("starts at quartiles:", [(s[order[i],0]-t0).item() for i in (0,255,256,511,512,767,768,1023)])
