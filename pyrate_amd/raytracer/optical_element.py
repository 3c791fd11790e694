"""``OpticalElement``: surfaces + materials + the per-surface trace loop with the
reference's bookkeeping (raytracer/optical_element.py:34-126, 324-379).  This loop is
the plugin-granular path (one propagate + one refract/reflect launch per surface,
ray-path forking for ``splitup``); all-isotropic sequences are normally traced by the
fused kernel through ``OpticalSystem.seqtrace``."""
from .localcoordinates import LocalCoordinatesTreeBase
from .ray import RayPath


class OpticalElement(LocalCoordinatesTreeBase):
    kind = "opticalelement"

    def __init__(self, lc, name=""):
        LocalCoordinatesTreeBase.__init__(self, lc, name=name)
        self.surfaces = {}
        self.materials = {}
        self.annotations["surf_mat_connection"] = {}

    @classmethod
    def p(cls, lc, name=""):
        return cls(lc, name=name)

    def addSurface(self, key, surface_object, materialkeys):
        """materialkeys: (material in minus-normal direction, material in plus-normal
        direction); keys missing from ``materials`` mean the background medium"""
        (minusNmat_key, plusNmat_key) = materialkeys
        if self.checkForRootConnection(surface_object.rootcoordinatesystem):
            self.surfaces[key] = surface_object
        else:
            raise Exception("surface coordinate system should be connected to OpticalElement "
                            "root coordinate system")
        self.annotations["surf_mat_connection"][key] = (minusNmat_key, plusNmat_key)

    def changeMaterialsForSurface(self, key, materialkeys):
        if key in self.annotations["surf_mat_connection"]:
            self.annotations["surf_mat_connection"][key] = tuple(materialkeys)

    def getSurfaces(self):
        return self.surfaces

    def getConnection(self, key):
        return self.annotations["surf_mat_connection"][key]

    def addMaterial(self, key, material_object, comment=""):
        if self.checkForRootConnection(material_object.lc):
            if key not in self.materials:
                self.materials[key] = material_object
                self.materials[key].comment = comment
        else:
            raise Exception("material coordinate system should be connected to OpticalElement "
                            "root coordinate system")

    def findoutWhichMaterial(self, mat1, mat2, current_mat):
        """material after refraction, by identity comparison (:109-126)"""
        return mat2 if (mat1 is current_mat) else mat1

    def _trace_plan(self, sequence, background_medium):
        """Resolve a surface sequence into trace steps before any ray is touched: for every
        entry the surface, the medium the rays travel in on their way to it, the medium that
        deflects them there and whether the deflection is a refraction.  A mirror leaves the
        rays in the medium they came in; a refracting surface hands them to whichever of its
        two media they are not in (identity comparison, like the reference)."""
        plan = []
        medium = background_medium
        for (surfkey, options) in sequence:
            surface = self.surfaces[surfkey]
            refracts = not options.get("is_mirror", False)
            arriving_in = medium
            if refracts:
                sides = [self.materials.get(key, background_medium)
                         for key in self.annotations["surf_mat_connection"][surfkey]]
                medium = self.findoutWhichMaterial(sides[0], sides[1], medium)
            plan.append((surface, arriving_in, medium, refracts))
        return plan

    def seqtrace(self, raybundle, sequence, background_medium, splitup=False):
        """Plugin-granular trace of one element: per step one propagate and one refract / reflect
        call on every open ray path; a deflection that returns several bundles (``splitup`` at a
        crystal interface) forks the path, forks are appended behind the existing paths."""
        paths = [RayPath(raybundle)]
        for (surface, arriving_in, deflecting, refracts) in self._trace_plan(sequence, background_medium):
            forks = []
            for path in paths:
                head = path.raybundles[-1]
                arriving_in.propagate(head, surface)
                deflect = deflecting.refract if refracts else deflecting.reflect
                (first, *others) = deflect(head, surface, splitup=splitup)
                for bundle in others:
                    fork = path.clone()
                    fork.appendRayBundle(bundle)
                    forks.append(fork)
                path.appendRayBundle(first)
            paths.extend(forks)
        return paths
