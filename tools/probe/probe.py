import ctypes, os, torch, time
t0=time.time()
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libprobe.so"))
dev = torch.device("cuda:0")
print("torch", torch.__version__, torch.version.hip, torch.cuda.get_device_name(0))
props = torch.cuda.get_device_properties(0)
print("CUs", props.multi_processor_count, "mem GB", props.total_memory/2**30)
s = torch.cuda.current_stream().cuda_stream
P = ctypes.c_void_p
x = torch.randn(1000, device=dev); y = torch.randn(1000, device=dev); y0 = y.clone()
rc = lib.probe_axpy(P(x.data_ptr()), P(y.data_ptr()), ctypes.c_float(2.0), 1000, P(s))
torch.cuda.synchronize(); print("axpy rc", rc, "err", (y - (2*x+y0)).abs().max().item())
# side stream
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    y1 = y0.clone()
    rc = lib.probe_axpy(P(x.data_ptr()), P(y1.data_ptr()), ctypes.c_float(3.0), 1000, P(st.cuda_stream))
st.synchronize(); print("axpy side-stream rc", rc, "err", (y1 - (3*x+y0)).abs().max().item())
A = torch.randn(32,16,device=dev).bfloat16().float(); B = torch.randn(16,32,device=dev).bfloat16().float()
C = torch.zeros(32,32,device=dev)
rc = lib.probe_mfma_bf16(P(A.data_ptr()),P(B.data_ptr()),P(C.data_ptr()),P(s)); torch.cuda.synchronize()
ref = (A.double()@B.double()).float()
print("mfma bf16 rc",rc,"err",(C-ref).abs().max().item(), "errT", (C.t()-ref).abs().max().item())
A = torch.randn(32,2,device=dev); B = torch.randn(2,32,device=dev); C = torch.zeros(32,32,device=dev)
rc = lib.probe_mfma_f32(P(A.data_ptr()),P(B.data_ptr()),P(C.data_ptr()),P(s)); torch.cuda.synchronize()
ref = (A.double()@B.double()).float()
print("mfma f32 rc",rc,"err",(C-ref).abs().max().item())
xx = torch.randn(512, device=dev); out = torch.zeros(256, dtype=torch.int32, device=dev)
rc = lib.probe_cvt(P(xx.data_ptr()), P(out.data_ptr()), 256, P(s)); torch.cuda.synchronize()
bf = xx.bfloat16().view(torch.int16).to(torch.int32) & 0xFFFF
exp = bf[0::2] | (bf[1::2] << 16)
print("cvt_pk rc", rc, "mismatch", ((out & 0xFFFFFFFF) != (exp & 0xFFFFFFFF)).sum().item())
# graph capture of a ctypes launch
g = torch.cuda.CUDAGraph()
y2 = y0.clone()
with torch.cuda.graph(g):
    lib.probe_axpy(P(x.data_ptr()), P(y2.data_ptr()), ctypes.c_float(1.0), 1000, P(torch.cuda.current_stream().cuda_stream))
g.replay(); g.replay(); torch.cuda.synchronize()
print("graph replay err", (y2-(y0+2*x)).abs().max().item())
print("cpu cores", os.cpu_count(), "elapsed", time.time()-t0)
