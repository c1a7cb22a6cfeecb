"""MI355X-native host mirror of the reference's ``mvn`` package for the volumetric-triangulation
forward path: same module paths, class names and call signatures
(``mvn.models.triangulation.VolumetricTriangulationNet`` ...), arithmetic in liblt_hip.so."""
