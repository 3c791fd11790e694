"""Pupil rasters -> (xpup, ypup) in the unit disk (reference: sampling2d/raster.py:37-164).
Deterministic input generators for the trace ("identical pupil samples"); host side."""
import math

import numpy as np


class RectGrid(object):
    def getGrid(self, nray):
        """square raster clipped to the unit disk; returns approximately nray points"""
        n_per_dim = int(round(math.sqrt(nray * 4.0 / math.pi)))
        dx = 1. / n_per_dim
        x1d = np.linspace(-1 + .25 * dx, 1 - .25 * dx, n_per_dim)
        (xpup, ypup) = np.meshgrid(x1d, x1d)
        xpup = np.reshape(xpup, n_per_dim ** 2)
        ypup = np.reshape(ypup, n_per_dim ** 2)
        ind = (xpup ** 2 + ypup ** 2) <= 1
        return (xpup[ind], ypup[ind])


class MeridionalFan(RectGrid):
    def getGrid(self, nray, phi=0.):
        """fan along the y axis (rotated by phi)"""
        rpup = np.linspace(-1, 1, nray)
        return (-rpup * math.sin(phi), rpup * math.cos(phi))


class SagittalFan(RectGrid):
    def getGrid(self, nray, phi=0.):
        return MeridionalFan().getGrid(nray, phi + math.pi / 2)


class ChiefAndComa(RectGrid):
    def getGrid(self, nray, phi=0.):
        """chief ray plus meridional and sagittal coma rays (5 points)"""
        xpup = np.array([0., 0., 0., 1., -1.])
        ypup = np.array([0., 1., -1., 0., 0.])
        return (xpup * math.cos(phi) - ypup * math.sin(phi), xpup * math.sin(phi) + ypup * math.cos(phi))


class Single(RectGrid):
    def __init__(self, xpup=0., ypup=0.):
        self.xpup = xpup
        self.ypup = ypup

    def getGrid(self, nray):
        return (np.array([self.xpup]), np.array([self.ypup]))
