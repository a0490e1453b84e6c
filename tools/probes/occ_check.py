import ctypes, os, sys, time, torch
ROOT="/root/repo"; sys.path.insert(0, ROOT)
from osu_diffusion_amd import _lib
occ = ctypes.CDLL(os.path.join(ROOT, "tools", "probes", "liboccupy.so"))
occ.occupy.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev=torch.device("cuda:0"); side=torch.cuda.Stream(device=dev); L=_lib.lib()
My,Nx,K=32768,3072,768
Y=torch.randn(My,K,device=dev).to(torch.bfloat16); X=torch.randn(Nx,K,device=dev).to(torch.bfloat16); out=torch.zeros(My,Nx,dtype=torch.bfloat16,device=dev); bias=torch.zeros(Nx,device=dev)
def go(): _lib.check(L.osud_op_gemm(0,_lib.EPI_BIAS_TE,_lib.ptr(Y),K,_lib.ptr(X),K,My,Nx,K,_lib.ptr(out),Nx,_lib.ptr(bias),None,0,0,0,None))
def t(n=20):
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): go()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1)*1e3/n
for _ in range(3): go()
torch.cuda.synchronize()
print("alone", t())
for usec in (40000, 400000, 1500000):
    torch.cuda.synchronize()
    t0=time.perf_counter()
    assert occ.occupy(8, usec, 96, ctypes.c_void_p(side.cuda_stream))==0
    time.sleep(0.005)
    a=t(); 
    time.sleep(0.05); b=t()
    side.synchronize(); el=time.perf_counter()-t0
    print(f"occupier {usec} us: gemm right after launch {a:.1f} us, 50 ms later {b:.1f} us; occupier really lasted {el*1e3:.0f} ms")
