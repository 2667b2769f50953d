"""metropolisengine_amd: MI355X-native many-chain Metropolis engine with the MetropolisEngine API surface.

``import metropolisengine_amd as me; me.MetropolisEngine(me.IsoQuadratic(1.0), initial_real_params=[0.0], temp=.01)``
mirrors ``import metropolisengine as me`` of the reference (README.md:13-17, metropolisengine/__init__.py:1).
The HIP library is loaded on first use of an engine; there is no CPU fallback.
"""
from .energy import (AbsReal0AtLeast, CylinderSurrogate, DenseQuadratic, DiagQuadratic, EnergySpec, IsoQuadratic,
                     LandauToy, RejectSpec, UserEnergy, UserReject)
from .engine import MetropolisEngine, set_cache_budget

__all__ = ["MetropolisEngine", "set_cache_budget", "EnergySpec", "IsoQuadratic", "DiagQuadratic", "DenseQuadratic", "LandauToy",
           "CylinderSurrogate", "UserEnergy", "RejectSpec", "AbsReal0AtLeast", "UserReject"]
__version__ = "0.1.0"
