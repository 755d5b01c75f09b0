"""rayopt_amd -- MI355X-native engine behind rayopt's GeometricTrace.propagate.

Public names mirror the reference's for the accelerated path
(``System, Element, Interface, Spheroid, Material, GeometricTrace,
system_from_yaml``).  The ray arithmetic runs in hand-written HIP kernels
(rayopt_amd/csrc) reached through the C ABI in include/rt_mi355.h; there is no
CPU implementation of the path in this package and no fallback.
"""
from .model import (System, Element, Interface, Spheroid, Pose, Material,
                    ConstantIndex, AbbeGlass, Conjugate, make_element)
from .formats import (system_from_yaml, system_from_json, system_from_dict,
                      system_to_yaml, system_to_json)
from .geometric_trace import GeometricTrace, Trace, DeviceRows
from .library import Library

FullTrace = GeometricTrace      # the reference's alias (geometric_trace.py:262)
from .engine import Engine, get_engine
from ._lib import EngineError
from . import prescriptions, bundles, pupil, merit, catalog

__all__ = [
    "System", "Element", "Interface", "Spheroid", "Pose", "Material",
    "ConstantIndex", "AbbeGlass", "Conjugate", "make_element",
    "system_from_yaml", "system_from_json", "system_from_dict",
    "system_to_yaml", "system_to_json", "GeometricTrace", "Trace",
    "DeviceRows", "FullTrace", "Library", "Engine", "get_engine",
    "EngineError", "prescriptions",
    "bundles", "pupil", "merit", "catalog",
]
