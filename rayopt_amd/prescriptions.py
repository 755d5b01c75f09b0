"""Benchmark / parity prescriptions (BASELINE.json
configs).  The reference ships only the Cooke-triplet geometry
(rayopt/test/test_raytrace.py:36-44); the others are data authored for this
project.  All materials are numeric so no glass catalogue is needed on the
GPU box.  The same YAML loads in the reference (``rayopt.system_from_yaml``)
and here (``rayopt_amd.system_from_yaml``)."""

SINGLET = """
description: 'C1: biconvex singlet, 4 surfaces, all spherical'
wavelengths: [587.56e-9]
object: {angle_deg: 5, pupil: {radius: 8.0}}
image: {type: finite, pupil: {radius: 0, update_radius: True}}
stop: 1
elements:
- {material: 1.0}
- {roc: 51.5, distance: 10.0, material: 1.5168, radius: 10.0}
- {roc: -51.5, distance: 5.0, material: 1.0, radius: 10.0}
- {distance: 48.2, radius: 8.0}
"""

# geometry of rayopt/test/test_raytrace.py:30-45 (the reference's only
# multi-surface geometric-trace fixture) with the catalogue glasses replaced
# by their numeric indices and the image radius opened from 0.364 to 20 so a
# clipped trace keeps finite directions at the image
COOKE = """
description: 'C2: oslo cooke triplet example 50mm f/4 20deg (reference test fixture), numeric indices'
wavelengths: [587.56e-9, 656.27e-9, 486.13e-9]
object: {angle_deg: 20, pupil: {radius: 6.25, aim: True}}
image: {type: finite, pupil: {radius: 0, update_radius: True}}
stop: 5
elements:
- {material: %(air)r}
- {roc: 21.25, distance: 5.0, material: %(sk16)r, radius: 6.5}
- {roc: -158.65, distance: 2.0, material: %(air)r, radius: 6.5}
- {roc: -20.25, distance: 6.0, material: %(f2)r, radius: 5.0}
- {roc: 19.6, distance: 1.0, material: %(air)r, radius: 5.0}
- {material: %(air)r, radius: 4.75}
- {roc: 141.25, distance: 6.0, material: %(sk16)r, radius: 6.5}
- {roc: -17.285, distance: 2.0, material: %(air)r, radius: 6.5}
- {distance: 42.95, radius: 20.}
"""

# per-wavelength (air, N-SK16, N-F2) index triples, read from the reference
# with its catalogue (tests/golden/make_golden.py prints them)
COOKE_INDICES = {
    587.56e-9: dict(air=1.0002771748755976, sk16=1.6204100608393477,
                    f2=1.6200532924653839),
    656.27e-9: dict(air=1.0002762521518107, sk16=1.6172717580453815,
                    f2=1.6150580546386029),
    486.13e-9: dict(air=1.000279356018886, sk16=1.6275566004521669,
                    f2=1.6320783029397603),
}


def cooke(wavelength=587.56e-9):
    return COOKE % COOKE_INDICES[wavelength]


DOUBLE_GAUSS = """
description: 'C3/C5: double-Gauss, 11 refracting surfaces + stop + image'
wavelengths: [587.56e-9]
object: {angle_deg: 14, pupil: {radius: 16.0}}
image: {type: finite, pupil: {radius: 0, update_radius: True}}
stop: 6
elements:
- {material: 1.0}
- {roc: 54.153, distance: 10.0, material: 1.60738, radius: 29.225}
- {roc: 152.522, distance: 8.747, material: 1.0, radius: 28.141}
- {roc: 35.951, distance: 0.5, material: 1.62041, radius: 24.296}
- {distance: 14.0, material: 1.60342, radius: 21.297}
- {roc: 22.270, distance: 3.777, material: 1.0, radius: 14.919}
- {distance: 14.253, material: 1.0, radius: 10.229}
- {roc: -25.685, distance: 12.428, material: 1.60342, radius: 13.188}
- {distance: 3.777, material: 1.62041, radius: 16.468}
- {roc: -36.980, distance: 10.834, material: 1.0, radius: 18.930}
- {roc: 196.417, distance: 0.5, material: 1.62041, radius: 21.311}
- {roc: -67.148, distance: 6.858, material: 1.0, radius: 21.646}
- {distance: 57.315, radius: 30.}
"""
DOUBLE_GAUSS_PUPIL_Z = 68.94    # entrance pupil distance from surface 0
DOUBLE_GAUSS_FIELD_DEG = 14.0

ASPHERE_PHONE = """
description: 'C4: 3-element even-asphere phone lens: stop + 6 aspheres + image'
wavelengths: [587.56e-9]
object: {angle_deg: 25, pupil: {radius: 0.6}}
image: {type: finite, pupil: {radius: 0, update_radius: True}}
stop: 1
elements:
- {material: 1.0}
- {distance: 0.5, material: 1.0, radius: 0.62}
- {roc: 1.35, conic: -0.5, aspherics: [0, -0.010, -0.020, 0.010], distance: 0.05, material: 1.5346, radius: 0.80}
- {roc: 5.20, conic: 0.0, aspherics: [0, 0.020, -0.040, 0.015], distance: 0.62, material: 1.0, radius: 0.85}
- {roc: -1.70, conic: 0.3, aspherics: [0, -0.080, 0.050, -0.020], distance: 0.48, material: 1.6142, radius: 0.95}
- {roc: -3.10, conic: 0.0, aspherics: [0, -0.060, 0.030, 0.004], distance: 0.36, material: 1.0, radius: 1.10}
- {roc: 2.10, conic: -2.0, aspherics: [0, -0.070, 0.012, -0.0012], distance: 0.34, material: 1.5346, radius: 1.55}
- {roc: 1.45, conic: -1.5, aspherics: [0, -0.060, 0.010, -0.0010], distance: 0.78, material: 1.0, radius: 1.85}
- {distance: 1.0, radius: 3.5}
"""

TORTURE = """
description: 'torture: tilts, decentres, conics, plane refraction, fold mirror, alternate intersection'
wavelengths: [587.56e-9]
object: {angle_deg: 2, pupil: {radius: 5.0}}
image: {type: finite, pupil: {radius: 0, update_radius: True}}
stop: 1
elements:
- {material: 1.0}
- {roc: 80, conic: -0.6, distance: 20, material: 1.5168, radius: 15, angles: [0.05, -0.03, 0.02]}
- {roc: -120, conic: 0.4, distance: 6, material: 1.0, radius: 15, direction: [0.02, -0.01, 1.0]}
- {distance: 10, material: 1.7, radius: 14, angles: [-0.1, 0.05, 0.0]}
- {distance: 3, material: 1.0, radius: 14}
- {roc: -200, conic: -1.3, distance: 40, material: mirror, radius: 25, angles: [0.03, 0.0, 0.0]}
- {roc: 60, distance: -25, material: 1.5168, radius: 14}
- {roc: 25, distance: -4, material: 1.0, radius: 12, alternate_intersection: true}
- {distance: -20, radius: 40}
"""

ALL = {"singlet": SINGLET, "cooke": cooke(), "double_gauss": DOUBLE_GAUSS,
       "asphere_phone": ASPHERE_PHONE, "torture": TORTURE}
