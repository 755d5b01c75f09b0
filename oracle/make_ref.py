"""TEST INFRASTRUCTURE ONLY -- recipe that carries the UNMODIFIED reference to
the GPU box.

``/root/reference`` exists in the build container and not on the GPU pool.
The reference is pure Python on this path (rayopt/geometric_trace.py,
system.py, elements.py ...; its two optional C/Cython extensions are not
needed and not built), so "building" it is: pack the package's ``*.py`` files,
byte for byte, from where they lie into ONE importable archive,
``oracle/_ref/rayopt_reference.zip`` (Python imports packages from a zip),
and put its glass database next to it -- an artefact like the built ``.so``
files: git-ignored (``.gitignore``: never in history), travels with the
gpurun snapshot (not in ``.gpurunignore``).  ``oracle/refshim.py`` imports it
from there when ``/root/reference`` is absent, which puts the reference itself -- not a port
-- beside the engine on the GPU box:

* ``bench.py``'s ``cpu_baseline`` leg times ``rayopt.GeometricTrace.propagate``
  (rayopt/geometric_trace.py:72-80) on the box's own host cores
  (``kind: "reference"``);
* ``tests/test_reference_live_gpu.py`` compares the device with the live
  reference on BASELINE configs C1, C2 and a C4 subsample;
* ``tests/test_dropin.py::test_reference_analysis_runs_on_the_gpu`` runs the
  reference's own ``Analysis`` (rayopt/analysis.py:76-143) on the device.

Nothing in ``rayopt_amd/`` imports it (tests/test_cabi.py pins that).

    python -m oracle.make_ref          # or __graft_entry__.build()
"""
import hashlib
import json
import os
import shutil
import zipfile

SOURCE = os.environ.get("RAYOPT_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
DEST = os.path.join(HERE, "_ref")
ARCHIVE = os.path.join(DEST, "rayopt_reference.zip")
LIBRARY_DB = os.path.join(DEST, "library.sqlite")
# what the path (and the Analysis consumer) imports; the test directory, the
# C / Cython sources and packaging files stay where they are
SKIP_DIRS = {"test", "__pycache__"}


def source_available():
    return os.path.isdir(os.path.join(SOURCE, "rayopt"))


def built():
    return os.path.isfile(ARCHIVE)


def _sources():
    src = os.path.join(SOURCE, "rayopt")
    for root, dirs, files in os.walk(src):
        dirs[:] = sorted(d for d in dirs if d not in SKIP_DIRS)
        for name in sorted(files):
            if name.endswith(".py"):
                path = os.path.join(root, name)
                yield os.path.relpath(path, SOURCE), path


def make(force=False):
    """Pack the reference package into ``oracle/_ref/`` (idempotent: an
    archive whose manifest matches the sources is left alone).  Returns the
    destination, or None when there is neither a source tree on this machine
    nor an archive that travelled here (the GPU box uses what travelled)."""
    if not source_available():
        return DEST if built() else None
    manifest = {}
    blobs = []
    for rel, path in _sources():
        with open(path, "rb") as f:
            data = f.read()
        manifest[rel] = hashlib.sha256(data).hexdigest()
        blobs.append((rel, data))
    note = os.path.join(DEST, "MANIFEST.json")
    if not force and built() and os.path.isfile(note) and \
            os.path.isfile(LIBRARY_DB):
        with open(note) as f:
            if json.load(f).get("sha256") == manifest:
                return DEST
    os.makedirs(DEST, exist_ok=True)
    tmp = ARCHIVE + ".tmp"
    with zipfile.ZipFile(tmp, "w", zipfile.ZIP_DEFLATED) as z:
        for rel, data in blobs:
            info = zipfile.ZipInfo(rel, date_time=(2020, 1, 1, 0, 0, 0))
            info.compress_type = zipfile.ZIP_DEFLATED
            z.writestr(info, data)
    os.replace(tmp, ARCHIVE)
    shutil.copyfile(os.path.join(SOURCE, "rayopt", "library.sqlite"),
                    LIBRARY_DB)
    for name in ("COPYING", "COPYING-GPL-3", "AUTHORS"):   # its licence
        a = os.path.join(SOURCE, name)
        if os.path.isfile(a):
            shutil.copyfile(a, os.path.join(DEST, name))
    with open(note, "w") as f:
        json.dump({"source": SOURCE, "sha256": manifest}, f, indent=1,
                  sort_keys=True)
    return DEST


if __name__ == "__main__":
    print(make())
