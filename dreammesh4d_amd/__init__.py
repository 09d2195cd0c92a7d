"""dreammesh4d_amd -- MI355X-native hot path for DreamMesh4D's dynamic stage.

See DESIGN.md.  The compute path is libdm4d_hip.so (hand-written HIP for gfx950,
C ABI declared in include/dm4d.h); this package is the Python host that mirrors
the reference's operator / plugin interface for that path.
"""
__version__ = "0.1.0"
