"""Developer aid: can the host CPU store directly into device memory (large BAR)? Each attempt runs in a child process
(a refused access is a SIGSEGV). Prints the store->visible latency seen by a polling kernel when it works."""
import ctypes, os, subprocess, sys

def child(kind):
    hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
    p = ctypes.c_void_p()
    if kind == "fine":
        rc = hip.hipExtMallocWithFlags(ctypes.byref(p), ctypes.c_size_t(4096), ctypes.c_uint(0x1))  # hipDeviceMallocFinegrained
    elif kind == "uncached":
        rc = hip.hipExtMallocWithFlags(ctypes.byref(p), ctypes.c_size_t(4096), ctypes.c_uint(0x3))  # hipDeviceMallocUncached
    else:
        rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(4096))
    print(kind, "alloc rc", rc, hex(p.value or 0), flush=True)
    if rc != 0:
        return
    hip.hipMemset(p, 0, ctypes.c_size_t(4096))
    hip.hipDeviceSynchronize()
    arr = (ctypes.c_uint32 * 16).from_address(p.value)
    arr[0] = 0x12345678  # host store into device memory
    v = arr[0]
    back = (ctypes.c_uint32 * 16)()
    hip.hipMemcpy(back, p, ctypes.c_size_t(64), ctypes.c_int(2))
    print(kind, "host store ok, host load 0x%x, device copy sees 0x%x" % (v, back[0]), flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for k in ("fine", "uncached", "plain"):
            r = subprocess.run([sys.executable, __file__, k], capture_output=True, text=True)
            print(r.stdout.strip(), "| exit", r.returncode, r.stderr.strip()[-200:])
