# forced 128x128 tile: 32x32x16 MFMAs (SDVAR_GEMM_H2_STAGES=4, default) against 16x16x32 (=6); checks the result against torch fp64 first
for v in 4 6; do echo "variant=$v"; SDVAR_GEMM_H2_STAGES=$v python - <<'PY'
import ctypes as C, torch, sys, os
sys.path.insert(0, os.getcwd())
from sdvar_amd import engine as E
lib=E.load_library(); dev=torch.device("cuda:0"); st=C.c_void_p(torch.cuda.current_stream().cuda_stream)
P=lambda t: C.c_void_p(t.data_ptr())
torch.manual_seed(0)
for (M,N,K,split) in [(200,384,256,1),(576,4096,1024,1),(1024,4096,1024,1),(1600,3072,1024,1),(2624,1024,4096,1),(4096,1024,4096,1),(4096,4096,1024,1),(6800,4096,1024,1),(6800,1024,4096,1)]:
    X=torch.randn(M,K,device=dev); W=torch.randn(N,K,device=dev)*0.02; b=torch.randn(N,device=dev); out=torch.empty(M,N,device=dev)
    Xp=torch.empty(2,M,K,dtype=torch.int16,device=dev); Wp=torch.empty(2,N,K,dtype=torch.int16,device=dev); wsc=torch.zeros(4,device=dev)
    E._check(lib.sdvar_op_split_planes_f16(P(X),P(Xp),M,K,M*K,None,st)); E._check(lib.sdvar_op_split_planes_f16(P(W),P(Wp),N,K,N*K,P(wsc),st))
    E._check(lib.sdvar_debug_set_gemm_cfg(128,split))
    run=lambda: E._check(lib.sdvar_op_gemm_f16x2(P(Xp),M*K,P(Wp),N*K,P(wsc),P(b),P(out),N,None,0,M,N,K,0,None,N,None,1,0,st))
    for _ in range(3): run()
    ref=(X.double()@W.double().T+b.double())
    err=((out.double()-ref).abs().max()/ref.abs().max()).item()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): run()
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)*1e3/30
    print(f"  M={M} N={N} K={K} split={split}: {us:.1f} us  {2.0*M*N*K/us/1e6:.0f} TF/s  max err/max {err:.2e}")
PY
done
