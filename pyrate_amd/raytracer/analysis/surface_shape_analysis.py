"""``ShapeAnalysis``: sag tables of a shape (reference: raytracer/analysis/surface_shape_analysis.py:
33-115).  The sag itself is evaluated on the GPU (``Shape.getSag`` -> prt_shape_eval); tables are
NumPy arrays / text files like the reference's.  Plotting is left to the caller."""
import numpy as np


class ShapeAnalysis(object):
    kind = "shapeanalysis"

    def __init__(self, shape, name=""):
        self.shape = shape
        self.name = name

    def generate_sag_matrices(self, xlinspace, ylinspace):
        """(xgrid, ygrid, zgrid) on the meshgrid of the two sample vectors"""
        (xgrid, ygrid) = np.meshgrid(xlinspace, ylinspace)
        zgrid = np.reshape(self.shape.getSag(xgrid.flatten(), ygrid.flatten()), np.shape(xgrid))
        return (xgrid, ygrid, zgrid)

    def generate_sag_table(self, xlinspace, ylinspace):
        """(3, n) table x, y, sag"""
        (xgrid, ygrid, zgrid) = self.generate_sag_matrices(xlinspace, ylinspace)
        return np.vstack((xgrid.flatten(), ygrid.flatten(), zgrid.flatten()))

    def load_sag_table(self, filename):
        return np.loadtxt(filename, dtype=float).T

    def save_sag_table(self, filename, xlinspace, ylinspace):
        np.savetxt(filename, self.generate_sag_table(xlinspace, ylinspace).T)

    def compare_with_sag_table(self, table):
        """sag of the shape minus the tabulated sag at the table's points"""
        return self.shape.getSag(table[0], table[1]) - table[2]
