"""Does a hipStreamWaitValue32 gate hold a stream (null stream and a pool stream) until hipStreamWriteValue32 opens it?  (tools/host_time_ranks.py relies on it.)"""
import ctypes, time, torch, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "automatic-speech-recognition_amd"))
hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
hip.hipStreamWaitValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint, ctypes.c_uint32]
hip.hipStreamWriteValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint]
hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
torch.zeros(1, device="cuda")
attr = ctypes.c_int(0)
hip.hipDeviceGetAttribute.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int]
sig = ctypes.c_void_p()
print("malloc", hip.hipExtMallocWithFlags(ctypes.byref(sig), 8, 0x2), hex(sig.value or 0))
print("memset", hip.hipMemset(sig, 0, 8))
for name, st in (("null", torch.cuda.current_stream()), ("pool", torch.cuda.Stream())):
    opener = torch.cuda.Stream(priority=-1)
    with torch.cuda.stream(st):
        rc = hip.hipStreamWaitValue32(ctypes.c_void_p(st.cuda_stream), sig, 5, 0, 0xFFFFFFFF)
        x = torch.ones(1024, device="cuda") * 2
        ev = torch.cuda.Event(); ev.record()
    time.sleep(0.3)
    held = not ev.query()
    rc2 = hip.hipStreamWriteValue32(ctypes.c_void_p(opener.cuda_stream), sig, 5, 0)
    t0 = time.perf_counter(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(name, "wait rc", rc, "held after 0.3 s:", held, "write rc", rc2, "sync after open %.3f ms" % (dt * 1e3), float(x[0]))
    hip.hipMemset(sig, 0, 8)
