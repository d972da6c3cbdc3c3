"""ppst_amd -- MI355X-native (gfx950) implementation of the PPST swap hot path.

Importing the package does not touch the GPU; the HIP library is loaded on first use of
``ppst_amd.ops`` and its absence is an error (there is no CPU fallback)."""
__version__ = "0.1.0"
