"""What hipModuleOccupancyMaxActiveBlocksPerMultiprocessor says about the kernels of a built librpt_hip.so (GPU box).
usage: python tools/occupancy_probe.py LIB.so NAME_SUBSTRING BLOCK_THREADS DYNAMIC_LDS_BYTES"""
import ctypes as C, glob, os, re, shutil, subprocess, sys, tempfile
lib, sub, threads, dyn = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
t = tempfile.mkdtemp(); shutil.copy(lib, os.path.join(t, "lib.so"))
subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", "lib.so"], cwd=t, capture_output=True)
hip = C.CDLL("libamdhip64.so")
assert hip.hipInit(0) == 0
for co in sorted(glob.glob(os.path.join(t, "lib.so.*gfx950*"))):
    syms = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-s", co], capture_output=True, text=True).stdout
    names = [l.split()[-1] for l in syms.split("\n") if " FUNC " in l and sub in l and not l.split()[-1].endswith(".kd")]
    if not names: continue
    mod = C.c_void_p()
    data = open(co, "rb").read()
    assert hip.hipModuleLoadData(C.byref(mod), data) == 0
    for n in names:
        f = C.c_void_p()
        if hip.hipModuleGetFunction(C.byref(f), mod, n.encode()) != 0: continue
        nb = C.c_int()
        rc = hip.hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(C.byref(nb), f, C.c_int(threads), C.c_size_t(dyn))
        print(os.path.basename(lib), n[:60], "rc", rc, "blocks per CU", nb.value)
shutil.rmtree(t)
