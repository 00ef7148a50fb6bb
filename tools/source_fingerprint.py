#!/usr/bin/env python3
"""sha256 over the device-side sources of librpt_hip.so (csrc/*.h, csrc/*.hip, in name order) and the Makefile (its compiler flags).

The GPU box receives a snapshot without .git, so a commit id is not available where bench.py runs; this fingerprint
identifies the kernel build instead.  PMC-derived figures kept under profiles/ carry it, and bench.py refuses to
report one measured on different kernel sources."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fingerprint():
    d = os.path.join(ROOT, "rust-path-tracer_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip")):
            h.update(name.encode())
            with open(os.path.join(d, name), "rb") as f:
                h.update(f.read())
    with open(os.path.join(ROOT, "Makefile"), "rb") as f:
        h.update(b"Makefile")
        h.update(f.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(fingerprint())
