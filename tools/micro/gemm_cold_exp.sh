# small-M GEMM with HBM-cold weights (a rotating set of weight tensors > the 256 MB Infinity Cache, as inside a model pass), slab launch alone (SDVAR_GEMM_DBG=8)
for old in 0 1; do echo "SDVAR_GEMM_SMALL_OLD=$old"; SDVAR_GEMM_DBG=8 SDVAR_GEMM_SMALL_OLD=$old python - <<'PY'
import ctypes as C, torch, sys, os
sys.path.insert(0, os.getcwd())
from sdvar_amd import engine as E
lib=E.load_library(); dev=torch.device("cuda:0"); st=C.c_void_p(torch.cuda.current_stream().cuda_stream)
P=lambda t: C.c_void_p(t.data_ptr())
torch.manual_seed(0)
shapes=[("qkv",3072,1024),("proj",1024,1024),("fc1",4096,1024),("fc2",1024,4096)]
for M in (16,64,144):
  for name,N,K in shapes:
    nW=max(2,int(600e6/(N*K*4)))
    X=torch.randn(M,K,device=dev); b=torch.randn(N,device=dev); out=torch.empty(M,N,device=dev)
    Xp=torch.empty(2,M,K,dtype=torch.int16,device=dev); wsc=torch.zeros(4,device=dev)
    E._check(lib.sdvar_op_split_planes_f16(P(X),P(Xp),M,K,M*K,None,st))
    W=torch.randn(N,K,device=dev)*0.02
    Wps=[torch.empty(2,N,K,dtype=torch.int16,device=dev) for _ in range(nW)]
    for w in Wps: E._check(lib.sdvar_op_split_planes_f16(P(W),P(w),N,K,N*K,P(wsc),st))
    res=[]
    for bm in (32,64):
      for split in (1,2,3,4,5,6,8,10,12,16,24,32):
        if split>K//32: continue
        E._check(lib.sdvar_debug_set_gemm_cfg(bm,split))
        def run(i): E._check(lib.sdvar_op_gemm_f16x2(P(Xp),M*K,P(Wps[i%nW]),N*K,P(wsc),P(b),P(out),N,None,0,M,N,K,0,None,N,None,1,0,st))
        for i in range(3): run(i)
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(2*nW): run(i)
        e1.record(); torch.cuda.synchronize()
        res.append((e0.elapsed_time(e1)*1e3/(2*nW),bm,split))
    res.sort()
    print(f"  M={M:4d} {name:4s}: "+", ".join(f"{t:.1f}us(bm{bm},s{sp})" for t,bm,sp in res[:5]), flush=True)
E._check(lib.sdvar_debug_set_gemm_cfg(0,0))
PY
done
