import torch, time
for (M,N,K) in ((262144,512,512),(786432,256,256),(262144,1024,1024)):
    x=torch.randn(M,K,device='cuda',dtype=torch.bfloat16); w=torch.randn(N,K,device='cuda',dtype=torch.bfloat16)*0.05
    for _ in range(3): y=x@w.t()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(20): y=x@w.t()
    b.record(); torch.cuda.synchronize()
    t=a.elapsed_time(b)/20/1e3
    print("hipBLASLt bf16 M %d N %d K %d: %.3f ms = %.0f TFLOP/s"%(M,N,K,t*1e3,2.0*M*N*K/t/1e12),flush=True)
