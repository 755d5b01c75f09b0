"""Seeded random prescriptions + ray bundles for differential testing.

Covers the whole surface zoo the path supports -- planes, spheres, conics on
both sides of the paraboloid, even aspheres (Newton), tilts, decentres,
mirrors with the reference's negative-distance convention, apertures that
vignette, alternate intersections, index changes incl. none -- in random
combinations no hand-written fixture would.
"""
import numpy as np


def random_prescription(seed):
    rng = np.random.default_rng(seed)
    nel = int(rng.integers(2, 9))
    n_obj = float(rng.choice([1.0, 1.0, 1.00027, 1.33]))
    elements = [{"material": n_obj}]
    sign = 1.            # flips after every mirror (negative distances)
    for j in range(nel):
        el = {}
        el["distance"] = sign*float(rng.uniform(0.5, 25.))
        radius = float(rng.uniform(3., 12.))
        el["radius"] = radius
        kind = rng.random()
        if kind > 0.25:
            roc = float(rng.uniform(1.6*radius, 250.))*float(rng.choice([-1, 1]))
            el["roc"] = roc
            if rng.random() < 0.5:
                k = float(rng.uniform(-2.5, 1.0))
                if k > -1 and radius**2 > 0.9/((1 + k)/roc**2):
                    k = -0.5
                el["conic"] = k
        if rng.random() < 0.25:
            nterm = int(rng.integers(1, 6))
            el["aspherics"] = [float(rng.normal()*3e-3/radius**(2*i + 1))
                               for i in range(nterm)]
        m = rng.random()
        if m < 0.12:
            el["material"] = "mirror"
        elif m < 0.85:
            el["material"] = float(rng.uniform(1.0, 1.9))
        if rng.random() < 0.3:
            el["angles"] = [float(a) for a in rng.uniform(-.06, .06, 3)]
        if rng.random() < 0.12:
            d = rng.uniform(-.03, .03, 2)
            el["direction"] = [float(d[0]), float(d[1]), sign*1.0]
            el["distance"] = abs(el["distance"])
        if rng.random() < 0.04 and "roc" in el:
            el["alternate_intersection"] = True
        elements.append(el)
        if el.get("material") == "mirror":
            sign = -sign
    elements.append({"distance": sign*float(rng.uniform(5., 60.)),
                     "radius": float(rng.uniform(5., 40.))})
    return {"wavelengths": [587.56e-9], "elements": elements}


def random_rays(seed, n, prescription):
    rng = np.random.default_rng(seed + 7919)
    rad = min(e["radius"] for e in prescription["elements"][1:-1])
    rad *= float(rng.uniform(0.5, 1.25))
    r = rad*np.sqrt(rng.random(n))
    phi = 2*np.pi*rng.random(n)
    y = np.zeros((n, 3))
    y[:, 0], y[:, 1] = r*np.cos(phi), r*np.sin(phi)
    ax, ay = np.radians(rng.uniform(-4, 4, 2))
    u = np.zeros((n, 3))
    u[:, 0] = np.sin(ax) + rng.normal(size=n)*2e-3
    u[:, 1] = np.sin(ay) + rng.normal(size=n)*2e-3
    u[:, 2] = np.sqrt(1 - u[:, 0]**2 - u[:, 1]**2)
    return y, u
