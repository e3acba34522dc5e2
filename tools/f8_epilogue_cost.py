import sys, os, statistics, torch
sys.path.insert(0, os.getcwd())
from composable_kernel_amd import capi
M=N=4096
dev=torch.device("cuda",0)
g=torch.Generator(device=dev); g.manual_seed(1)
sp=torch.cuda.current_stream().cuda_stream
for K in (128, 4096):
    a=(torch.randint(8,120,(M,K),generator=g,device=dev,dtype=torch.int16)+128*torch.randint(0,2,(M,K),generator=g,device=dev,dtype=torch.int16)).to(torch.uint8)
    q=torch.randint(-8,8,(N,K),generator=g,device=dev,dtype=torch.int8)
    b=torch.empty((N,K//2),dtype=torch.uint8,device=dev); capi.pack_int4(q.data_ptr(),K,K,N,b.data_ptr(),K//2,sp)
    b2=torch.empty_like(b); capi.preshuffle_int4(b.data_ptr(),K//2,K,N,b2.data_ptr(),K,sp)
    d0=torch.rand((N,),generator=g,device=dev); d1=torch.rand((M,),generator=g,device=dev)
    c=torch.empty((M,N),dtype=torch.float16,device=dev)
    probs={}
    for name,epi in (("mm",capi.EPI_MULTIPLY_MULTIPLY),("mm_gelu",capi.EPI_MULTIPLY_MULTIPLY_FASTGELU)):
        probs[name]=capi.fpAintB_problem(a.data_ptr(),b2.data_ptr(),None,c.data_ptr(),M,N,K,a_dtype=capi.F8_E4M3FN,b_dtype=capi.I4_PRESHUFFLED,c_dtype=capi.F16,epilogue=epi,d0=d0.data_ptr(),d1=d1.data_ptr())
    capi.run_timed(probs["mm"],None,sp,0,1000)
    t={k:[] for k in probs}
    for _ in range(7):
        for k,p in probs.items(): t[k].append(capi.run_timed(p,None,sp,0,50))
    print(K,{k:round(statistics.median(v)*1e3,2) for k,v in t.items()})
