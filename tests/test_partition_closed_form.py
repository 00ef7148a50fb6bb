"""The closed form of the reference's in-place two-pointer partition (src/bvh.rs:281-292) that the GPU BVH build uses
(rust-path-tracer_amd/csrc/k_bvh_build.h, passes B1 / F / B2) against the sequential loop itself, on random and on
exhaustive small inputs.  CPU only: this pins the DERIVATION; tests/test_gpu_bvh_build.py pins the kernels."""
import itertools
import random


def sequential(cls):
    """bvh.rs:281-292 on positions 0..n-1; cls[i] = centroid(i) < split.  Returns (final order of original ids, nl)."""
    arr = list(range(len(cls)))
    a, b = 0, len(arr) - 1
    while a <= b:
        if cls[arr[a]]:
            a += 1
        else:
            arr[a], arr[b] = arr[b], arr[a]
            b -= 1
    return arr, a


def closed_form(cls):
    n = len(cls)
    first, last = 0, n - 1
    nl = sum(cls)
    dest = [None] * n
    back = range(last, first + nl - 1, -1)                     # suffix, descending (pass B1)
    H = sum(1 for p in range(first, first + nl) if not cls[p])
    rb_at_l = [0] * (H + 1)
    m = rb = 0
    seen = {}
    for q in back:
        seen[q] = (m, rb)
        if cls[q]:
            rb_at_l[m] = rb
            m += 1
        else:
            rb += 1
    assert m == H
    base_rb = rb_at_l[H - 1] if H >= 1 else 0
    hole_pos, i = [], 0
    for p in range(first, first + nl):                         # prefix, ascending (pass F)
        if cls[p]:
            dest[p] = p
        else:
            dest[p] = last - (i + (rb_at_l[i - 1] if i >= 1 else 0))
            hole_pos.append(p)
            i += 1
    for q in back:                                             # suffix again (pass B2)
        m, rb = seen[q]
        if cls[q]:
            dest[q] = hole_pos[m]
        elif m == H:                                           # the tail below the lowest suffix left-side element
            dest[q] = last - (H + base_rb if q == first + nl else H + rb + 1)
        else:
            dest[q] = last - ((m + 1) + rb)
    out = [None] * n
    for p in range(n):
        assert out[dest[p]] is None
        out[dest[p]] = p
    return out, nl


def test_closed_form_equals_the_loop_exhaustively_up_to_12_elements():
    for n in range(0, 13):
        for bits in itertools.product((False, True), repeat=n):
            cls = list(bits)
            assert closed_form(cls) == sequential(cls), cls


def test_closed_form_equals_the_loop_on_random_inputs():
    rnd = random.Random(7)
    for _ in range(20000):
        n = rnd.randint(1, 200)
        p = rnd.random()
        cls = [rnd.random() < p for _ in range(n)]
        assert closed_form(cls) == sequential(cls)
