"""MI355X-native MCMC-SLAM proposal-evaluation path.

`slam_amd.capi` binds the C ABI of libmcmcgpu.so (HIP kernels for gfx950);
`slam_amd.synthetic` generates problems of the benchmark's shape;
`slam_amd/host/` holds the C++ drop-in for the reference's mcmc_slam class.
"""
from . import capi, synthetic  # noqa: F401
