#!/usr/bin/env python3
"""Generate tests/golden/math_cr_vectors.npz: inputs and CORRECTLY ROUNDED float32 results of the
transcendentals rpt_math.h provides, computed with mpmath at 100 digits (one rounding, ties to even).

These are data vectors (inputs + expected outputs); the reference itself holds no vectors for its
libm calls.  Run:  python tools/gen_math_golden.py
"""
import os

import mpmath as mp
import numpy as np

mp.mp.dps = 100
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def round_f32(v):
    """Nearest float32 (ties to even) of an mpf, denormals and overflow included."""
    if mp.isnan(v):
        return np.float32(np.nan)
    if v == 0:
        return np.float32(0.0)
    if mp.isinf(v):
        return np.float32(np.inf if v > 0 else -np.inf)
    _, e = mp.frexp(abs(v))                   # |v| = m * 2^e, 0.5 <= m < 1
    q_exp = max(int(e) - 24, -149)
    n = mp.nint(v / mp.ldexp(1, q_exp))       # ties to even
    r = float(mp.ldexp(n, q_exp))
    if abs(r) >= 2.0 ** 128:
        return np.float32(np.inf if r > 0 else -np.inf)
    return np.float32(r)


def main():
    rng = np.random.default_rng(20260102)
    N = 3000
    f32 = np.float32
    out = {}

    def add(name, xs, fn, ys=None):
        xs = np.asarray(xs, f32)
        if ys is None:
            res = np.array([round_f32(fn(mp.mpf(float(x)))) for x in xs], f32)
            out[name + "_x"] = xs
        else:
            ys = np.asarray(ys, f32)
            res = np.array([round_f32(fn(mp.mpf(float(x)), mp.mpf(float(y)))) for x, y in zip(xs, ys)], f32)
            out[name + "_x"] = xs
            out[name + "_y"] = ys
        out[name + "_r"] = res

    ang = np.concatenate([rng.uniform(0, 2 * np.pi, N), rng.uniform(-50, 50, N // 2), rng.uniform(0, np.pi / 2, N // 2),
                          [0.0, 1e-20, 1e-6, np.pi / 2, np.pi, 2 * np.pi, 100.0, -100.0, 12345.678]])
    add("sin", ang, mp.sin)
    add("cos", ang, mp.cos)
    unit = np.concatenate([rng.uniform(-1, 1, N), np.sqrt(rng.uniform(0, 1, N)), 1 - rng.uniform(0, 1e-4, 200),
                           [0.0, 1.0, -1.0, 0.5, -0.5, 0.49999997, 0.50000006, 1 - 2.0 ** -24, 1e-8]])
    add("acos", unit, mp.acos)
    add("asin", unit, mp.asin)
    ex = np.concatenate([rng.uniform(-30, 5, N), rng.uniform(-104, 89, N // 2), -rng.uniform(0, 1e-3, 200),
                         [0.0, 1.0, -1.0, 88.7, -87.4, -100.0, -103.9]])
    add("exp", ex, mp.exp)
    px = np.concatenate([rng.uniform(0, 4, N), rng.uniform(0, 1e-3, 300), rng.uniform(0.05, 3.2, N)])
    py = np.concatenate([np.full(N, 2.2), rng.uniform(0.1, 3, 300), np.full(N, 1.5)])
    add("pow", px, lambda a, b: mp.power(a, b) if a != 0 else mp.mpf(0), py)
    ay = rng.uniform(-3, 3, N)
    ax = rng.uniform(-3, 3, N)
    add("atan2", ay, mp.atan2, ax)
    path = os.path.join(ROOT, "tests", "golden", "math_cr_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items() if k.endswith("_r")})


if __name__ == "__main__":
    main()
