"""``OpticalSystemAnalysis``: bundle generation + trace convenience with the reference's
signatures (raytracer/analysis/optical_system_analysis.py:44-191).  For the isotropic
background medium the dispersion relation gives k = n * unitvector directly (the reference
runs a per-ray scipy.linalg.eig for the same result, material/material.py:456-499); E is
*a* unit vector perpendicular to k, like the reference's eigenvector."""
import math

import numpy as np

from ... import engine
from ...sampling2d.raster import RectGrid, device_tables_of
from ..globalconstants import degree, standard_wavelength
from ..ray import RayBundle, default_device


def _perp_field(k):
    """unit E perpendicular to each k on the device (prt_efield_perp), back as numpy"""
    kd = engine.to_device_rays(k, default_device())
    return engine.efield_perp(kd).cpu().numpy()


class OpticalSystemAnalysis(object):
    kind = "opticalsystemanalysis"

    def __init__(self, os, seq, name=""):
        self.opticalsystem = os
        self.sequence = seq
        self.name = name
        self.initial_bundles = None

    def _background_index(self, wave):
        mat = self.opticalsystem.material_background
        if not hasattr(mat, "get_optical_index"):
            raise Exception("bundle generation needs an isotropic background medium")
        return float(mat.get_optical_index(np.zeros((3, 1)), wave))

    def collimated_bundle(self, nrays, properties_dict=None, wave=standard_wavelength):
        """(:83-122) keys: startx, starty, startz, raster, radius, anglex, angley"""
        pd = properties_dict or {}
        (startx, starty, startz) = (pd.get("startx", 0.), pd.get("starty", 0.), pd.get("startz", 0.))
        rasterobj = pd.get("raster", RectGrid())
        radius = pd.get("radius", 1.0)
        (angley, anglex) = (pd.get("angley", 0.0), pd.get("anglex", 0.0))
        (px, py) = rasterobj.getGrid(nrays)
        origin = np.vstack((radius * px + startx, radius * py + starty, startz * np.ones_like(px)))
        unit = np.zeros_like(origin)
        unit[0, :] = math.sin(angley) * math.cos(anglex)
        unit[1, :] = math.sin(anglex)
        unit[2, :] = math.cos(angley) * math.cos(anglex)
        k = self._background_index(wave) * unit
        return (origin, k, _perp_field(k))

    def divergent_bundle(self, nrays, properties_dict=None, wave=standard_wavelength):
        """(:124-165) keys: startx, starty, startz, raster, radius (half cone angle), anglex, angley"""
        pd = properties_dict or {}
        (startx, starty, startz) = (pd.get("startx", 0.), pd.get("starty", 0.), pd.get("startz", 0.))
        rasterobj = pd.get("raster", RectGrid())
        radius = pd.get("radius", 45.0 * degree)
        (angley, anglex) = (pd.get("angley", 0.0), pd.get("anglex", 0.0))
        (ax, ay) = rasterobj.getGrid(nrays)
        origin = np.vstack((startx * np.ones_like(ax), starty * np.ones_like(ax), startz * np.ones_like(ax)))
        unit = np.zeros_like(origin)
        unit[0, :] = np.sin(angley + radius * ax) * np.cos(anglex + radius * ay)
        unit[1, :] = np.sin(anglex + radius * ay)
        unit[2, :] = np.cos(angley + radius * ax) * np.cos(anglex + radius * ay)
        k = self._background_index(wave) * unit
        return (origin, k, _perp_field(k))

    def aim(self, numrays, rays_dict, bundletype="collimated", wave=standard_wavelength):
        """(:167-181).  For the deterministic rasters (RectGrid, HexGrid, the fans, CircularGrid -- those
        with ``device_tables``) both bundle types are generated directly on the GPU (prt_raster_bundle:
        the pupil samples are the reference's bit for bit, nothing is uploaded); random rasters and
        hand-picked rays are sampled on the host."""
        rays_dict = rays_dict or {}
        if bundletype not in ("collimated", "divergent"):
            raise KeyError(bundletype)
        rasterobj = rays_dict.get("raster", RectGrid())
        tables = device_tables_of(rasterobj, numrays)
        if tables is not None:
            self.initial_bundles = [self._bundle_on_device(tables, bundletype, rays_dict, wave)]
            return
        call = {"collimated": self.collimated_bundle, "divergent": self.divergent_bundle}
        (o, k, e) = call[bundletype](numrays, rays_dict, wave=wave)
        self.initial_bundles = [RayBundle(x0=o, k0=k, Efield0=e, wave=wave)]

    def _bundle_on_device(self, tables, bundletype, pd, wave):
        (angley, anglex) = (pd.get("angley", 0.0), pd.get("anglex", 0.0))
        index = self._background_index(wave)
        start = (pd.get("startx", 0.), pd.get("starty", 0.), pd.get("startz", 0.))
        dev = default_device()
        if bundletype == "divergent":
            (x, k, e, _) = engine.raster_bundle_device(tables, "divergent", dev, radius=pd.get("radius", 45.0 * degree),
                                                       start=start, anglex=anglex, angley=angley, index=index)
            return RayBundle(x0=x, k0=k, Efield0=e, wave=wave, device=dev)
        unit = np.array([math.sin(angley) * math.cos(anglex), math.sin(anglex),
                         math.cos(angley) * math.cos(anglex)])
        kvec = index * unit
        # one unit vector perpendicular to k for the whole bundle (same rule as prt_efield_perp)
        axis = np.eye(3)[int(np.argmin(np.abs(kvec)))] if abs(kvec[1]) > min(abs(kvec[0]), abs(kvec[2])) \
            else np.array([0., 1., 0.])
        evec = np.cross(kvec, axis)
        evec = evec / np.linalg.norm(evec)
        # k and E are one vector each for the whole bundle: nothing is stored per ray and the fused trace
        # loads only the origins (prt_trace_ex, uniform first segment); RayBundle.k / .Efield still return
        # the reference's (P,3,N) arrays
        (x, uni, _, _) = engine.raster_bundle_device(tables, "collimated", dev, radius=pd.get("radius", 1.0),
                                                     start=start, kvec=kvec, evec=evec, uniform=True)
        return RayBundle(x0=x, k0=None, Efield0=None, wave=wave, device=dev, uniform=uni)

    def trace(self, **kwargs):
        return [self.opticalsystem.seqtrace(ib, self.sequence, **kwargs) for ib in self.initial_bundles]

    def set_sequence(self, seq):
        self.sequence = seq

    def get_sequence(self):
        return self.sequence

    def get_footprint(self):
        """placeholder of the reference (:272-281)"""
        return np.array([0, 0])

    def get_matrices(self, **kwargs):
        raise NotImplementedError()

    def prettyprint(self):
        for (elem, elemseq) in self.sequence:
            print(elem)
            for (surf, opts) in elemseq:
                print("    " + surf + " " + str(opts))

    # ---- convenience wrappers (:193-303); results leave the device only here -----------------
    def trace_3d_global(self, x0, k0, wave=standard_wavelength, **kwargs):
        """trace from given start points / wave vectors (E = ey); per field point, per ray path:
        [(x, k) at the start of every bundle], global coordinates"""
        self.initial_bundles = [RayBundle(x0=x0, k0=k0, Efield0=None, wave=wave)]
        return [[[(rb.x[0], rb.k[0]) for rb in rp.raybundles] for rp in fp] for fp in self.trace(**kwargs)]

    def _flat_surfaces(self):
        out = []
        for (elem, elemseq) in self.sequence:
            out += [self.opticalsystem.elements[elem].surfaces[surf] for (surf, _) in elemseq]
        return out

    def trace_3d_local(self, **kwargs):
        """as trace_3d_global, each (x, k) pair in the frame of the surface it is paired with
        (zip of the flattened sequence with the path's bundles, like the reference)"""
        surfs = self._flat_surfaces()
        return [[[(sf.rootcoordinatesystem.returnGlobalToLocalPoints(X),
                   sf.rootcoordinatesystem.returnGlobalToLocalDirections(K))
                  for (sf, (X, K)) in zip(surfs, rp)] for rp in fp]
                for fp in self.trace_3d_global(**kwargs)]

    def trace_2d_local(self, **kwargs):
        """footprints: the x, y components of trace_3d_local"""
        return [[[(X[:2], K[:2]) for (X, K) in rp] for rp in fp] for fp in self.trace_3d_local(**kwargs)]

    def get_spot(self, raypath):
        """(image-plane points in the last surface's frame (2, N), RMS spot radius about the centroid)"""
        from .ray_analysis import RayBundleAnalysis
        last_surf = self._flat_surfaces()[-1]
        last_bundle = raypath.raybundles[-1]
        local = last_surf.rootcoordinatesystem.returnGlobalToLocalPoints(last_bundle.x[-1])
        return (local[0:2, :], RayBundleAnalysis(last_bundle).get_rms_spot_size_centroid())
