import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def rss():
    d = {}
    for l in open("/proc/self/status"):
        if l.startswith(("VmRSS", "RssAnon", "RssShmem")):
            k, v = l.split(":"); d[k] = int(v.split()[0]) // 1024
    return d
hip = ctypes.CDLL("libamdhip64.so")
print("start", rss(), flush=True)
n = ctypes.c_int(); hip.hipGetDeviceCount(ctypes.byref(n)); hip.hipSetDevice(0)
p = ctypes.c_void_p(); hip.hipMalloc(ctypes.byref(p), 1 << 20)
print("hip init + 1 MiB malloc", rss(), flush=True)
ss = []
for i in range(8):
    s = ctypes.c_void_p(); hip.hipStreamCreateWithFlags(ctypes.byref(s), 1); ss.append(s)
    hip.hipMemsetAsync(p, 0, 1 << 20, s); hip.hipStreamSynchronize(s)
    print("stream", i + 1, rss(), flush=True)
from curdleproofs_pie_amd import _native as N
print("library", rss(), flush=True)
cs = []
for i in range(4):
    cs.append(N.Context(0)); print("context", i + 1, rss(), flush=True)
for i, c in enumerate(cs):
    c.gen_scalars_device(c.alloc(32 << 10), 1 << 10, 5); c.sync()
    print("after a kernel on context", i + 1, rss(), flush=True)
