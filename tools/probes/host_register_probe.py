"""How long do hipHostRegister / hipHostUnregister of one 960 x 960 x 3 page take, and how fast is H2D from pageable vs registered vs pinned memory?"""
import ctypes, time
import numpy as np
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipHostRegister.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
hip.hipHostUnregister.argtypes = [ctypes.c_void_p]
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
hip.hipHostMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
n = 960 * 960 * 3
pages = [np.random.randint(0, 255, n, dtype=np.uint8) for _ in range(16)]
d = ctypes.c_void_p()
assert hip.hipMalloc(ctypes.byref(d), n * 16) == 0
hip.hipDeviceSynchronize()
def t(f, reps=1):
    t0 = time.perf_counter(); [f() for _ in range(reps)]; return (time.perf_counter() - t0) / reps
for _ in range(2):
    tp = t(lambda: [hip.hipMemcpy(d.value + i * n, p.ctypes.data, n, 1) for i, p in enumerate(pages)])
print(f"pageable H2D: {tp / 16 * 1e6:.0f} us per page = {n * 16 / tp / 1e9:.1f} GB/s")
treg = t(lambda: [hip.hipHostRegister(p.ctypes.data, n, 0) for p in pages])
tc = t(lambda: [hip.hipMemcpy(d.value + i * n, p.ctypes.data, n, 1) for i, p in enumerate(pages)])
tun = t(lambda: [hip.hipHostUnregister(p.ctypes.data) for p in pages])
print(f"register {treg / 16 * 1e6:.0f} us  copy {tc / 16 * 1e6:.0f} us ({n * 16 / tc / 1e9:.1f} GB/s)  unregister {tun / 16 * 1e6:.0f} us per page")
h = ctypes.c_void_p()
assert hip.hipHostMalloc(ctypes.byref(h), n * 16, 0) == 0
buf = (ctypes.c_uint8 * (n * 16)).from_address(h.value)
dst = np.frombuffer(buf, dtype=np.uint8)
tm = t(lambda: [np.copyto(dst[i * n:(i + 1) * n], p) for i, p in enumerate(pages)])
tpin = t(lambda: hip.hipMemcpy(d.value, h.value, n * 16, 1))
print(f"memcpy to pinned {tm / 16 * 1e6:.0f} us per page ({n * 16 / tm / 1e9:.1f} GB/s, one thread)  pinned H2D {n * 16 / tpin / 1e9:.1f} GB/s")
