"""Spectral lines [mm] and canonical axes (reference: raytracer/globalconstants.py:32-58)."""
import math

import numpy as np

iline = 0.3650E-3
hline = 0.4047E-3
gline = 0.4358E-3
Fprimeline = 0.4800E-3
Fline = 0.4861E-3
eline = 0.5461E-3
dline = 0.5876E-3
Dline = 0.5893E-3
Cprimeline = 0.6438E-3
Cline = 0.6563E-3
rline = 0.7065E-3
sline = 0.8521E-3
tline = 1.0140E-3

standard_wavelength = dline

canonical_ex = np.array([1, 0, 0])
canonical_ey = np.array([0, 1, 0])
canonical_ez = np.array([0, 0, 1])

degree = math.pi / 180.0
numerical_tolerance = 1e-17
