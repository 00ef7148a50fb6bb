"""Randomised check that the shared deterministic math (rpt_math.h, host build) returns the CORRECTLY ROUNDED float on\nN random arguments per function, against mpmath at 60 digits.  usage: python tools/cr_check.py [N]   (CPU only)"""
import importlib, sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tools'))
import numpy as np, mpmath as mp
from gen_math_golden import round_f32
hip = importlib.import_module("rust-path-tracer_amd.hip")
mp.mp.dps = 60
rng = np.random.default_rng(777)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
f32 = np.float32
def check(name, op, xs, fn, ys=None):
    xs = np.asarray(xs, f32)
    got = hip.debug_math_host(op, xs, None if ys is None else np.asarray(ys, f32))
    bad = 0
    t = time.time()
    for i in range(len(xs)):
        ref = round_f32(fn(mp.mpf(float(xs[i]))) if ys is None else fn(mp.mpf(float(xs[i])), mp.mpf(float(f32(ys[i])))))
        if ref.view(np.uint32) != got[i].view(np.uint32) and not (ref == 0 and got[i] == 0):
            bad += 1
            if bad <= 3: print("  MISROUND", name, xs[i], None if ys is None else ys[i], got[i], ref)
    print(f"{name}: {len(xs)} inputs, {bad} not correctly rounded ({time.time()-t:.0f} s)")
check("sin", 0, rng.uniform(0, 2*np.pi, N), mp.sin)
check("cos", 1, rng.uniform(0, 2*np.pi, N), mp.cos)
check("acos", 2, np.sqrt(rng.uniform(0, 1, N)), mp.acos)
check("exp", 3, rng.uniform(-40, 3, N), mp.exp)
check("pow 2.2", 4, rng.uniform(0, 4, N), lambda a, b: mp.power(a, b) if a != 0 else mp.mpf(0), np.full(N, 2.2))
check("pow 1.5", 4, rng.uniform(0.05, 3.2, N // 2), lambda a, b: mp.power(a, b), np.full(N // 2, 1.5))
check("asin", 5, rng.uniform(-1, 1, N // 2), mp.asin)
check("atan2", 6, rng.uniform(-3, 3, N // 2), mp.atan2, rng.uniform(-3, 3, N // 2))
