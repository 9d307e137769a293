"""Kernel specs shared by the oracle tests and the GPU parity tests.

Mirrors the kernel list of the reference's test/kernels.jl:209-245 restricted
to the families on the hot path: every stationary leaf,
Noise, Const, Sum/Prod composites, Masked, Fixed.  d = 3 like the reference.
"""
import math

D = 3
LL3 = [0.3, -0.2, 0.5]

LEAVES = [
    ("se_iso", 0.3, 0.1),
    ("se_ard", LL3, 0.1),
    ("mat12_iso", 0.2, -0.1),
    ("mat12_ard", LL3, -0.1),
    ("mat32_iso", 0.2, 0.4),
    ("mat32_ard", LL3, 0.4),
    ("mat52_iso", -0.1, 0.2),
    ("mat52_ard", LL3, 0.2),
    ("rq_iso", 0.3, 0.2, 0.4),
    ("rq_ard", LL3, 0.2, 0.4),
    ("noise", -0.5),
    ("const", 0.3),
]

COMPOSITES = [
    ("sum", ("se_iso", 0.3, 0.1), ("rq_iso", 0.3, 0.2, 0.4)),
    ("prod", ("se_iso", 0.3, 0.1), ("mat12_iso", 0.2, -0.1)),
    ("sum", ("sum", ("se_ard", LL3, 0.0), ("mat52_iso", math.log(0.7), math.log(0.5))), ("noise", math.log(0.05))),
    ("prod", ("sum", ("se_iso", 0.3, 0.3), ("mat12_iso", 0.3, 0.3)), ("rq_iso", 0.3, 0.3, 0.3)),
    ("masked", ("se_iso", 0.3, 0.3), [0]),
    ("sum", ("masked", ("se_iso", 0.3, 0.3), [0]), ("masked", ("rq_ard", [0.1, 0.2], 0.3, 0.3), [1, 2])),
    ("fixed", ("se_iso", 0.3, 0.3), [0]),
    ("prod", ("const", 0.2), ("masked", ("mat32_ard", [0.4, -0.3], 0.0), [2, 0])),
]

ALL = LEAVES + COMPOSITES


def ids(cases):
    def name(s):
        if s[0] in ("sum", "prod"):
            return f"{s[0]}({name(s[1])},{name(s[2])})"
        if s[0] in ("masked", "fixed"):
            return f"{s[0]}({name(s[1])})"
        return s[0]

    return [name(c) for c in cases]
