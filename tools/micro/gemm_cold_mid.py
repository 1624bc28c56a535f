#!/usr/bin/env python3
"""Mid-M GEMM with HBM-cold weights (rotating weight tensors, 600 MB per shape), every (row tile, K split) candidate; SDVAR_GEMM_DBG=8 times the slab launch alone.
python tools/micro/gemm_cold_mid.py [M ...]"""
import ctypes as C, torch, sys, os
sys.path.insert(0, os.getcwd())
from sdvar_amd import engine as E
lib=E.load_library(); dev=torch.device("cuda:0"); st=C.c_void_p(torch.cuda.current_stream().cuda_stream)
P=lambda t: C.c_void_p(t.data_ptr())
torch.manual_seed(0)
shapes=[("qkv",3072,1024,0),("proj",1024,1024,2),("fc1",4096,1024,1),("fc2",1024,4096,2)]
Ms=[int(a) for a in sys.argv[1:]] or [256,400,576,1024]
for M in Ms:
  for name,N,K,epi in shapes:
    nW=max(2,int(600e6/(N*K*4)))
    X=torch.randn(M,K,device=dev); b=torch.randn(N,device=dev); out=torch.empty(M,N,device=dev); gate=torch.randn(16,6*1024,device=dev)
    Xp=torch.empty(2,M,K,dtype=torch.int16,device=dev); wsc=torch.zeros(4,device=dev); outp=torch.empty(2,M,N,dtype=torch.int16,device=dev)
    E._check(lib.sdvar_op_split_planes_f16(P(X),P(Xp),M,K,M*K,None,st))
    W=torch.randn(N,K,device=dev)*0.02
    Wps=[torch.empty(2,N,K,dtype=torch.int16,device=dev) for _ in range(nW)]
    for w in Wps: E._check(lib.sdvar_op_split_planes_f16(P(W),P(w),N,K,N*K,P(wsc),st))
    def run(i): E._check(lib.sdvar_op_gemm_f16x2(P(Xp),M*K,P(Wps[i%nW]),N*K,P(wsc),P(b),P(out),N,P(outp),M*N,M,N,K,epi,P(out) if epi==2 else None,N,P(gate) if epi==2 else None,M//16,6*1024,st))
    def timeit():
        for i in range(3): run(i)
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(2*nW): run(i)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1)*1e3/(2*nW)
    E._check(lib.sdvar_debug_set_gemm_cfg(0,0)); auto=timeit()
    res=[]
    for bm in (32,64,128,256):
      for split in (1,2,3,4,5,6,8,10,12,16):
        if split>K//64: continue
        E._check(lib.sdvar_debug_set_gemm_cfg(bm,split)); res.append((timeit(),bm,split))
    res.sort()
    print(f"  M={M:4d} {name:4s}: auto {auto:5.1f}us | "+", ".join(f"{t:.1f}us(bm{bm},s{sp})" for t,bm,sp in res[:5])+f" | mfma-ideal {2.0*M*N*K*3/1.5e15*1e6:.1f}us", flush=True)
E._check(lib.sdvar_debug_set_gemm_cfg(0,0))
