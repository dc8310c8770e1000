import sys, torch, ctypes as C
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from swapping_autoencoder_pytorch_amd import hip_lib as L
lib=L.get(); dev='cuda:0'
def t(m,n,k):
    a=torch.randn(m,k,device=dev); b=torch.randn(n,k,device=dev); c=torch.empty(m,n,device=dev)
    st=torch.cuda.current_stream().cuda_stream
    fn=lambda: lib.call("gemm_f32", a.data_ptr(), b.data_ptr(), None, c.data_ptr(), m,n,k, k,1, 1,k, n, 1.0, st)
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    ref=a@b.t()
    print(m,n,k,"%.1f us"%(e0.elapsed_time(e1)/20*1e3), "maxerr %.2e"%((c-ref).abs().max().item()/ref.abs().max().item()))
for mnk in [(16,2048,2048),(16,512,2048),(8,512,2048),(128,2048,3072),(16,1,512),(16,512,8192)]: t(*mnk)
