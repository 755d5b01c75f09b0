"""Prescription I/O: YAML/JSON dict <-> System (schema of rayopt/formats.py:
``distance`` is the distance *to* the surface from the previous vertex,
``roc`` the radius of curvature, ``radius`` the clear semi-aperture, a numeric
``material`` a constant refractive index)."""
import json

import yaml

from .model import System


def system_from_dict(dat):
    dat = dict(dat)
    if dat.pop("type", "system") != "system":
        raise ValueError("not a system prescription")
    return System(**dat)


def system_from_yaml(text):
    return system_from_dict(yaml.safe_load(text))


def system_from_json(text):
    return system_from_dict(json.loads(text))


def system_to_yaml(system):
    return yaml.safe_dump(system.dict())


def system_to_json(system):
    return json.dumps(system.dict())
