"""Does a hipMemsetAsync issued while a stream is being captured become a node that runs on every replay?"""
import ctypes, torch
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
for nbytes in (4, 264, 4096, 1 << 20):
    t = torch.full((nbytes,), 5, dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            rc = hip.hipMemsetAsync(t.data_ptr(), 0, nbytes, torch.cuda.current_stream().cuda_stream)
            t.add_(1)          # a kernel node behind it
    torch.cuda.synchronize()
    after_capture = int(t[0])
    res = []
    for i in range(3):
        t.fill_(9); torch.cuda.synchronize()
        g.replay(); torch.cuda.synchronize()
        res.append(int(t[0]))
    print(f"bytes {nbytes:8d}: rc {rc}, value after capture {after_capture} (5 = nothing ran), after replays {res} (1 = memset then +1; 10 = memset missing)")
