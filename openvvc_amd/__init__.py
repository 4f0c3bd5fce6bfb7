"""MI355X-native reconstruction back-end for OpenVVC (gfx950 HIP kernels behind a C ABI).

`capi`   ctypes binding of include/ovvc_hip.h (libovvc_hip.so, the product)
`engine` thin Python host harness over the engine entry points
`synth`  synthetic "recorded picture" generator used by tests and bench.py
"""
__all__ = ["capi", "engine", "synth"]
