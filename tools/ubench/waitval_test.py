import ctypes, torch, time
hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda:0")
flag = torch.zeros(1, dtype=torch.int32, device=dev)
a, b = torch.cuda.Stream(), torch.cuda.Stream()
x = torch.zeros(1 << 22, device=dev)
hip.hipStreamWaitValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint, ctypes.c_uint32]
hip.hipStreamWaitValue32.restype = ctypes.c_int
out = torch.zeros(1, device=dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
rc = hip.hipStreamWaitValue32(ctypes.c_void_p(b.cuda_stream), ctypes.c_void_p(flag.data_ptr()), 5, 0, 0xFFFFFFFF)  # 0 = GTE
print("rc", rc)
with torch.cuda.stream(b):
    out.add_(1.0)
ev = torch.cuda.Event(); ev.record(b)
time.sleep(0.2)
print("b done before flag?", ev.query())
with torch.cuda.stream(a):
    for _ in range(50): x.add_(1.0)
    flag.fill_(7)
torch.cuda.synchronize()
print("b done after flag?", ev.query(), float(out), time.perf_counter() - t0)
# capture test: can the wait sit in front of a graph replay?
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.graph(g, stream=s):
    out.add_(1.0)
flag.zero_(); torch.cuda.synchronize()
rc = hip.hipStreamWaitValue32(ctypes.c_void_p(b.cuda_stream), ctypes.c_void_p(flag.data_ptr()), 3, 0, 0xFFFFFFFF)
with torch.cuda.stream(b):
    g.replay()
ev = torch.cuda.Event(); ev.record(b)
time.sleep(0.1)
print("graph ran before flag?", ev.query())
flag.fill_(3); torch.cuda.synchronize()
print("graph ran after flag?", ev.query(), float(out))
