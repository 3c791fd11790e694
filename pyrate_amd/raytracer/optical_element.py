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

    def seqtrace(self, raybundle, sequence, background_medium, splitup=False):
        current_material = background_medium
        rpaths = [RayPath(raybundle)]
        for (surfkey, surfoptions) in sequence:
            refract_flag = not surfoptions.get("is_mirror", False)
            rpaths_new = []
            current_surface = self.surfaces[surfkey]
            (mnmat, pnmat) = self.annotations["surf_mat_connection"][surfkey]
            mnmat = self.materials.get(mnmat, background_medium)
            pnmat = self.materials.get(pnmat, background_medium)
            for rp in rpaths:
                current_material.propagate(rp.raybundles[-1], current_surface)
            if refract_flag:
                current_material = self.findoutWhichMaterial(mnmat, pnmat, current_material)
            deflect = current_material.refract if refract_flag else current_material.reflect
            for rp in rpaths:
                raybundles = deflect(rp.raybundles[-1], current_surface, splitup=splitup)
                for rb in raybundles[1:]:
                    rpathprime = rp.clone()
                    rpathprime.appendRayBundle(rb)
                    rpaths_new.append(rpathprime)
                rp.appendRayBundle(raybundles[0])
            rpaths = rpaths + rpaths_new
        return rpaths
