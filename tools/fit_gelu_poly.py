#!/usr/bin/env python3
"""Fit of safevla_amd/asmgen/gelu_poly.py's coefficients (run once; prints the fp32 bit patterns to paste into COEF_BITS)."""
import struct
import numpy as np
from numpy.polynomial import chebyshev as C
from scipy.special import erf
c, n = 4.0, 8
t = np.linspace(1e-7, c, 200001)
phi = 0.5 * erf(t / np.sqrt(2))
z = t * t / 8 - 1
A = C.chebvander(z, n) * t[:, None]
coef, *_ = np.linalg.lstsq(A, phi, rcond=None)
mono = C.cheb2poly(coef)
bits = [struct.unpack("<I", struct.pack("<f", float(v)))[0] for v in mono]
print("COEF_BITS = [" + ", ".join(f"0x{b:08x}" for b in bits) + "]")
print("max |t P - (Phi - 1/2)| in fp64:", np.abs(A @ coef - phi).max())
