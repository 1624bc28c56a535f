# small-M GEMM: the deep-ring LDS-DMA kernel against the register-staged one (SDVAR_GEMM_SMALL_OLD=1), forced 32- / 64-row tiles, best K split each;
# every timed configuration is first checked against torch fp64
for old in 0 1; do echo "SDVAR_GEMM_SMALL_OLD=$old (SDVAR_GEMM_DBG=${SDVAR_GEMM_DBG:-0}: 8 = slab launch alone, no reduce, unchecked)"; SDVAR_GEMM_SMALL_OLD=$old python - <<'PY'
import ctypes as C, torch, sys, os
sys.path.insert(0, os.getcwd())
from sdvar_amd import engine as E
lib=E.load_library(); dev=torch.device("cuda:0"); st=C.c_void_p(torch.cuda.current_stream().cuda_stream)
P=lambda t: C.c_void_p(t.data_ptr())
torch.manual_seed(0)
shapes=[("qkv",3072,1024,0),("proj",1024,1024,0),("fc1",4096,1024,0),("fc2",1024,4096,0)]
for M in (16,64,144):
  for name,N,K,epi in shapes:
    X=torch.randn(M,K,device=dev); W=torch.randn(N,K,device=dev)*0.02; b=torch.randn(N,device=dev); out=torch.empty(M,N,device=dev)
    Xp=torch.empty(2,M,K,dtype=torch.int16,device=dev); Wp=torch.empty(2,N,K,dtype=torch.int16,device=dev); wsc=torch.zeros(4,device=dev)
    E._check(lib.sdvar_op_split_planes_f16(P(X),P(Xp),M,K,M*K,None,st)); E._check(lib.sdvar_op_split_planes_f16(P(W),P(Wp),N,K,N*K,P(wsc),st))
    ref=(X.double()@W.double().T+b.double())
    res=[]
    for bm in (32,64):
      for split in (1,2,3,4,5,6,8,10,12,16):
        if split>K//64: continue
        E._check(lib.sdvar_debug_set_gemm_cfg(bm,split))
        run=lambda: E._check(lib.sdvar_op_gemm_f16x2(P(Xp),M*K,P(Wp),N*K,P(wsc),P(b),P(out),N,None,0,M,N,K,0,None,N,None,1,0,st))
        out.zero_(); run(); torch.cuda.synchronize()
        err=((out.double()-ref).abs().max()/ref.abs().max()).item()
        assert err<5e-6 or os.environ.get('SDVAR_GEMM_DBG'),(M,name,bm,split,err)
        for _ in range(3): run()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): run()
        e1.record(); torch.cuda.synchronize()
        res.append((e0.elapsed_time(e1)*1e3/30,bm,split))
    res.sort()
    print(f"  M={M:4d} {name:4s}: "+", ".join(f"{t:.1f}us(bm{bm},s{sp})" for t,bm,sp in res[:4]), flush=True)
E._check(lib.sdvar_debug_set_gemm_cfg(0,0))
PY
done
