"""arap_flow_amd -- MI355X-native (HIP, gfx950) ARAP optical-flow hot path.

Only what the path needs: csrc/ (HIP kernels + C ABI, built into lib/libarapopt.so), the ctypes
binding (capi), the host-side mirror of the reference's driver layer (opt), .flo I/O (flo) and the
synthetic DAVIS-shaped input generator used by the bench and the tests (synth).
"""
__version__ = "0.1.0"
