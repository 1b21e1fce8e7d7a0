"""emfusion_amd -- MI355X-native (gfx950) implementation of EM-Fusion's per-frame volumetric hot path.

The product is the C-ABI shared library ``libemf_hip.so`` (include/emf_hip.h, sources in
emfusion_amd/csrc) plus the C++ host classes that keep the reference's ``emf::TSDF`` /
``emf::ObjTSDF`` / ``emf::EMFusion`` surface.  This Python package is only the harness side:
a ctypes binding used by tests/ and bench.py, with PyTorch-ROCm providing device memory, streams
and torch.distributed.  There is no CPU fallback: importing :mod:`emfusion_amd.ops` without the
built library raises.
"""

__version__ = "0.1.0"
