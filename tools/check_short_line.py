#!/usr/bin/env python3
"""Exhaustive check of the quotient of rrt_device.h's short_line_cell: for every segment shorter than 64 steps
m_k = floor((2 minor k + major) / (2 major)) equals (int)(float(num) * rcp + 0.5 rcp) with the reciprocal off by up to 3 ulp,
fused multiply-add or not.  (The integer closed form is include/rrt_line.h:61-79, which the oracle uses.)"""
import numpy as np

bad = 0
for major in range(0, 64):
    den = 2 * major
    d = np.float32(den if major > 0 else 1)
    for ulp in (-3, -1, 0, 1, 3):
        rcp = np.float32((np.float32(1.0) / d) * (1 + ulp * 2.0**-23))
        half = np.float32(np.float32(0.5) * rcp)
        for minor in range(0, major + 1):
            k = np.arange(0, major + 1)
            num = 2 * minor * k + major
            exact = (num // den) if den > 0 else np.zeros_like(num)
            unfused = (num.astype(np.float32) * rcp + half).astype(np.float32)
            fused = (num.astype(np.float64) * np.float64(rcp) + np.float64(half)).astype(np.float32)
            for q in (unfused, fused):
                bad += int((q.astype(np.int32) != exact).sum())
print("mismatches:", bad)
raise SystemExit(1 if bad else 0)
