"""Host-side mirror of the reference's raytracer interface (same class / method /
argument names, same error behaviour) on top of the HIP engine.  See optical_system.py."""
