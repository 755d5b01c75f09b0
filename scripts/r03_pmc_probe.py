"""Workload for the PMC passes of r03_pmc.sh: the headline batch traced with
resident_lds = the default (two workgroups per CU), then 0 (seven), a few
launches each; the dispatch order tells the passes apart (first N default,
next N uncapped)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rayopt_amd as ra
from rayopt_amd import prescriptions as P
from bench import workload_rays

n = 10_000_000
system = ra.system_from_yaml(P.DOUBLE_GAUSS)
y, u = workload_rays(n, 0)
g = ra.GeometricTrace(system)
g.rays_given(y, u)
for lds in (-1, 0):
    g.engine.set_option("resident_lds", lds)
    for _ in range(6):
        g.propagate(clip=True)
g.engine.sync()
