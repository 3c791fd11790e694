"""Pupil rasters -> (xpup, ypup) in the unit disk (reference: sampling2d/raster.py:37-164).
Input generators for the trace ("identical pupil samples").  ``getGrid`` is the host form with the
reference's signature.  The deterministic rasters are outer products of 1-d tables, and for those
``device_tables`` returns the tables from which the GPU builds the same samples bit for bit
(``prt_raster_bundle``: point (i, j) = (xa[j] * xb[i], ya[j] * yb[i]), optionally clipped to the unit
disk) -- no sample array is ever uploaded.  Angles ``phi`` are in degrees like in the reference."""
import math

import numpy as np


def device_tables_of(rasterobj, nray):
    """``rasterobj.device_tables(nray)`` if those tables describe the samples ``rasterobj.getGrid`` returns, else
    None (host sampling, like the reference, which always calls getGrid: analysis/optical_system_analysis.py:
    99-100).  The tables belong to the class that defines them: a subclass that overrides ``getGrid`` WITHOUT
    overriding ``device_tables`` has its own samples and inherits somebody else's tables -- not used."""
    if not hasattr(rasterobj, "device_tables"):
        return None
    mro = type(rasterobj).__mro__

    def owner(name):
        for (depth, k) in enumerate(mro):
            if name in k.__dict__:
                return depth
        return len(mro)
    if owner("device_tables") > owner("getGrid"):      # getGrid was overridden further down than the tables
        return None
    return rasterobj.device_tables(nray)


def _inside_unit_disk(x, y):
    keep = (x ** 2 + y ** 2) <= 1
    return (x[keep], y[keep])


class RectGrid(object):
    def device_tables(self, nray):
        """[(xa, xb, ya, yb, clip), ...]: the raster as outer products x = xa[j] * xb[i],
        y = ya[j] * yb[i] (i slow, j fast), one entry per sub-raster in output order; None for
        rasters that are not of this form (random ones, hand-picked rays)"""
        x1d = self._samples_1d(nray)
        one = np.ones_like(x1d)
        return [(x1d, one, one, x1d, True)]

    @staticmethod
    def _samples_1d(nray):
        n_per_dim = int(round(math.sqrt(nray * 4.0 / math.pi)))
        dx = 1. / n_per_dim
        return np.linspace(-1 + .25 * dx, 1 - .25 * dx, n_per_dim)

    def getGrid(self, nray):
        """square raster clipped to the unit disk; returns approximately nray points"""
        x1d = self._samples_1d(nray)
        (xpup, ypup) = np.meshgrid(x1d, x1d)
        return _inside_unit_disk(np.reshape(xpup, x1d.shape[0] ** 2), np.reshape(ypup, x1d.shape[0] ** 2))


class HexGrid(RectGrid):
    def getGrid(self, nray):
        """hexagonal raster = two interleaved rectangular lattices (lattice + basis), each clipped
        to the disk and stacked one after the other"""
        (x1d, y1d) = self._lattice_1d(nray)
        nx = x1d.shape[0]
        (xa, ya) = np.meshgrid(x1d, y1d)
        xb = xa + 0.5 * (x1d[1] - x1d[0])
        yb = ya + 0.5 * (y1d[1] - y1d[0])
        (xa, ya) = _inside_unit_disk(xa.reshape(nx ** 2), ya.reshape(nx ** 2))
        (xb, yb) = _inside_unit_disk(xb.reshape(nx ** 2), yb.reshape(nx ** 2))
        return (np.hstack((xa, xb)), np.hstack((ya, yb)))

    @staticmethod
    def _lattice_1d(nray):
        nx = int(round(math.sqrt(2 * math.sqrt(3) * nray / math.pi) + 1))
        x1d = np.linspace(-1, 1, nx)
        return (x1d, x1d * math.sqrt(3))

    def device_tables(self, nray):
        (x1d, y1d) = self._lattice_1d(nray)
        one = np.ones_like(x1d)
        # the shifted lattice: the same element-wise additions the host form does on the mesh
        (x2, y2) = (x1d + 0.5 * (x1d[1] - x1d[0]), y1d + 0.5 * (y1d[1] - y1d[0]))
        return [(x1d, one, one, y1d, True), (x2, one, one, y2, True)]


class RandomGrid(RectGrid):
    def device_tables(self, nray):
        return None

    def getGrid(self, nray):
        """uniformly random points of the square, clipped to the disk (global NumPy generator,
        x drawn before y, like the reference)"""
        nsquare = int(round(nray * 4.0 / math.pi))
        xpup = 2. * np.random.random(nsquare) - 1.
        ypup = 2. * np.random.random(nsquare) - 1.
        return _inside_unit_disk(xpup, ypup)


class PoissonDiskSampling(RectGrid):
    def device_tables(self, nray):
        return None

    def getGrid(self, nray, tries=30):
        """blue-noise points with a minimum mutual distance of 1/sqrt(4 nray / pi) (Bridson's
        algorithm on [-1, 1]^2, clipped to the disk).  The reference uses its own Poisson2D dart
        thrower on the same square with the same distance; both are random, so only the point
        statistics agree."""
        n_per_dim = int(round(math.sqrt(nray * 4.0 / math.pi)))
        r = 1. / n_per_dim
        cell = r / math.sqrt(2.)
        ncell = int(math.ceil(2.0 / cell))
        grid = -np.ones((ncell, ncell), dtype=int)
        pts = []
        active = []

        def put(p):
            pts.append(p)
            active.append(len(pts) - 1)
            grid[int(p[0] / cell), int(p[1] / cell)] = len(pts) - 1

        put(np.random.random(2) * 2.0)
        while active:
            pick = active[np.random.randint(len(active))]
            base = pts[pick]
            for _ in range(tries):
                rad = r * (1. + np.random.random())
                ang = 2. * math.pi * np.random.random()
                cand = base + rad * np.array([math.cos(ang), math.sin(ang)])
                if not (0. <= cand[0] < 2. and 0. <= cand[1] < 2.):
                    continue
                (ci, cj) = (int(cand[0] / cell), int(cand[1] / cell))
                near = grid[max(ci - 2, 0):ci + 3, max(cj - 2, 0):cj + 3].ravel()
                if all(np.sum((pts[q] - cand) ** 2) >= r * r for q in near if q >= 0):
                    put(cand)
                    break
            else:
                active.remove(pick)
        sample = np.array(pts) - 1.0
        return _inside_unit_disk(sample[:, 0], sample[:, 1])


class MeridionalFan(RectGrid):
    def getGrid(self, nray, phi=0.):
        """fan along the y axis, rotated by phi [deg] towards -x"""
        t = np.linspace(-1, 1, nray)
        alpha = phi / 180. * math.pi
        return (t * -math.sin(alpha), t * math.cos(alpha))

    def device_tables(self, nray, phi=0.):
        alpha = phi / 180. * math.pi
        t = np.linspace(-1, 1, nray)
        return [(np.array([-math.sin(alpha)]), t, np.array([math.cos(alpha)]), t, False)]


class SagitalFan(RectGrid):
    """(spelling of the reference)"""

    def getGrid(self, nray, phi=0.):
        return MeridionalFan().getGrid(nray, phi - 90.)

    def device_tables(self, nray, phi=0.):
        return MeridionalFan().device_tables(nray, phi - 90.)


SagittalFan = SagitalFan


class ChiefAndComa(RectGrid):
    def device_tables(self, nray):
        return None

    def getGrid(self, nray, phi=0.):
        """chief ray (twice) and the four marginal rays of the meridional / sagittal sections"""
        alpha = phi / 180. * math.pi
        (sa, ca) = (math.sin(alpha), math.cos(alpha))
        return (np.array([0, 0, -sa, sa, ca, -ca], dtype=float), np.array([0, 0, ca, -ca, sa, -sa], dtype=float))


class Single(RectGrid):
    def __init__(self, xpup=0., ypup=0.):
        self.xpup = xpup
        self.ypup = ypup

    def device_tables(self, nray):
        return None

    def getGrid(self, nray, xpup=None, ypup=None):
        return (np.array([self.xpup if xpup is None else xpup]),
                np.array([self.ypup if ypup is None else ypup]))


class CircularGrid(RectGrid):
    def getGrid(self, nray, requidistant=True):
        """polar raster: sqrt(nray) radii x sqrt(nray) azimuths; requidistant=False spaces the radii
        for nearly equal area elements"""
        (r, phi) = self._polar_1d(nray, requidistant)
        (rr, pp) = np.meshgrid(r, phi)
        return ((rr * np.cos(pp)).flatten(), (rr * np.sin(pp)).flatten())

    @staticmethod
    def _polar_1d(nray, requidistant):
        n = int(round(math.sqrt(nray)))
        r = np.linspace(0, 1, num=n)
        if not requidistant:
            r = np.sqrt(r)
        return (r, np.linspace(0, 2. * math.pi, num=n, endpoint=False))

    def device_tables(self, nray, requidistant=True):
        (r, phi) = self._polar_1d(nray, requidistant)
        return [(r, np.cos(phi), r, np.sin(phi), False)]
