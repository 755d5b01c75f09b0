"""The kernel's arithmetic header, compiled for the host, against the golden
vectors of the reference -- at the tolerances the GPU parity tests use.  This
is how the HIP code's numerics are checked in a container without a GPU; the
`-m gpu` tests repeat the same comparison through the C ABI on the device."""
import numpy as np
import pytest

import rayopt_amd as ra
from rayopt_amd.pack import pack_system, resolve_range

from conftest import golden_names, load_golden, assert_parity, case_rtol


@pytest.mark.parametrize("rays_per_lane", [1, 2, 4])
@pytest.mark.parametrize("name", golden_names())
def test_kernel_math_matches_reference(hostemu, name, rays_per_lane):
    g = load_golden(name)
    system = ra.system_from_yaml(g["yaml"])
    a, b = resolve_range(len(system), g["start"], g["stop"])
    table, ns = pack_system(system, g["l"],
                            system.refractive_index(g["l"], 0), a, b)
    Y, U, I, T = hostemu(table, g["y0"], g["u0"], a, b, g["clip"],
                         rays_per_lane)
    rtol = case_rtol(g)
    for label, got, want in (("y", Y, g["y"]), ("u", U, g["u"]),
                             ("i", I, g["i"]), ("t", T, g["t"])):
        assert_parity(got, want[a:b], rtol, "%s.%s" % (name, label))
        # and in fact bit for bit: closed-form surfaces, tilted elements,
        # the Newton solve of the aspheres
        assert np.array_equal(got, want[a:b], equal_nan=True), (name, label)
