#!/usr/bin/env python
"""What a helper call costs through the C ABI (ab_phymbl): a scalar specific = a one-cell host array (staging, launch, copy back, synchronise),
and 100 000 host cells.  GPU box: python tools/phymbl_call_latency.py"""
import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np, ctypes as C
import aerobulk_amd as ab
from aerobulk_amd import _lib
lib = _lib.load()
x = np.array([290.0]); p = np.array([101000.0]); y = np.zeros(1)
pin = (C.c_void_p * 2)(x.ctypes.data, p.ctypes.data); pout = (C.c_void_p * 1)(y.ctypes.data); par = (C.c_double * 2)(0., 0.)
for _ in range(50): lib.ab_phymbl(17, 1, pin, 2, pout, 1, par, 0, 0, None, None)
t0 = time.perf_counter()
N = 2000
for _ in range(N): lib.ab_phymbl(17, 1, pin, 2, pout, 1, par, 0, 0, None, None)
dt = (time.perf_counter() - t0) / N
print(f"q_sat scalar (1-cell host arrays) through ab_phymbl: {dt*1e6:.1f} us per call, value {y[0]:.12e}")
xs = np.full(100000, 290.0); ps = np.full(100000, 101000.0); ys = np.zeros(100000)
pin = (C.c_void_p * 2)(xs.ctypes.data, ps.ctypes.data); pout = (C.c_void_p * 1)(ys.ctypes.data)
for _ in range(5): lib.ab_phymbl(17, 100000, pin, 2, pout, 1, par, 0, 0, None, None)
t0 = time.perf_counter()
for _ in range(50): lib.ab_phymbl(17, 100000, pin, 2, pout, 1, par, 0, 0, None, None)
print(f"q_sat on 100 000 host cells: {(time.perf_counter()-t0)/50*1e6:.1f} us per call")
