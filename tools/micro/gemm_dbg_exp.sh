# needs a timing-experiment build of the library: make -C sdvar_amd/csrc clean all EXTRA=-DSDVAR_TIMING_EXPERIMENTS (the product build has no such switches)
for d in 0 1 2 4 5 7; do echo "dbg=$d"; SDVAR_GEMM_DBG=$d python - <<'PY'
import ctypes as C, torch, sys, os
sys.path.insert(0, os.getcwd())
from sdvar_amd import engine as E
lib=E.load_library(); dev=torch.device("cuda:0"); st=C.c_void_p(torch.cuda.current_stream().cuda_stream)
P=lambda t: C.c_void_p(t.data_ptr())
for (M,N,K,epi) in [(4096,4096,1024,0),(4096,1024,4096,0),(1024,3072,1024,0)]:
    X=torch.randn(M,K,device=dev); W=torch.randn(N,K,device=dev)*0.02; b=torch.randn(N,device=dev); out=torch.empty(M,N,device=dev)
    Xp=torch.empty(2,M,K,dtype=torch.int16,device=dev); Wp=torch.empty(2,N,K,dtype=torch.int16,device=dev); wsc=torch.zeros(4,device=dev)
    E._check(lib.sdvar_op_split_planes_f16(P(X),P(Xp),M,K,M*K,None,st)); E._check(lib.sdvar_op_split_planes_f16(P(W),P(Wp),N,K,N*K,P(wsc),st))
    E._check(lib.sdvar_debug_set_gemm_cfg(128,1))
    run=lambda: E._check(lib.sdvar_op_gemm_f16x2(P(Xp),M*K,P(Wp),N*K,P(wsc),P(b),P(out),N,None,0,M,N,K,epi,None,N,None,1,0,st))
    for _ in range(3): run()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)*1e3/20
    print(f"  v2 M={M} N={N} K={K}: {us:.1f} us  {2.0*M*N*K/us*1e-6:.0f} TF/s")
PY
done
