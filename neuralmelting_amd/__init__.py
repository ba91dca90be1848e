"""neuralmelting_amd — MI355X-native NPT-HMC + replica-exchange sampler.

Drop-in for the data-parallel hot path of walkernr/neuralMelting's scripts/lammps_remcmc.py
(gen_samples -> gen_mc_params -> replica_exchange, remcmc:977-995): hand-written HIP kernels for
gfx950 behind a C-ABI (include/nm.h), a ctypes binding (`engine`) and a driver that keeps the
reference's command-line flags and file formats (`remcmc`).
"""
from .engine import Engine, NMError  # noqa: F401

__all__ = ['Engine', 'NMError']
