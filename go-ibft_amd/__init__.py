"""go-ibft_amd — MI355X (gfx950) batch verifier behind go-ibft's Verifier hooks.

The product is ``csrc/libibftgpu.so`` (hand-written HIP kernels + the C ABI of
``include/ibftgpu.h``).  This Python package is only the host-side mirror used by
tests and bench.py: ``build`` compiles the library, ``verifier`` binds it with ctypes.
There is no CPU fallback: importing works anywhere, but creating a ``BatchVerifier``
raises unless the HIP library is built and a gfx950 device is present.
"""
__all__ = ["build", "verifier"]
