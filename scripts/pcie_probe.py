"""End-to-end host<->device rates of the boundary (not part of `value`)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rayopt_amd as ra
from bench import workload_rays

n = 10_000_000
system = ra.system_from_yaml(ra.prescriptions.DOUBLE_GAUSS)
y, u = workload_rays(n, 0)
g = ra.GeometricTrace(system)
for rep in range(3):
    t0 = time.perf_counter(); g.rays_given(y, u); t1 = time.perf_counter()
    print("rays_given %d rays (480 MB AoS): %.3f s -> %.2f GB/s" % (n, t1 - t0, 0.48/(t1 - t0)))
g.propagate(clip=True)
for rep in range(3):
    g.y.invalidate(12, 13)
    t0 = time.perf_counter(); a = g.y[-1]; t1 = time.perf_counter()
    print("download y[-1] (240 MB): %.3f s -> %.2f GB/s" % (t1 - t0, 0.24/(t1 - t0)))
t0 = time.perf_counter(); r = g.rms(i=1); t1 = time.perf_counter()
print("device rms: %.4f s" % (t1 - t0))
t0 = time.perf_counter(); x, yy, t = g.opd(radius=100., resample=0); t1 = time.perf_counter()
print("device opd rays (+240 MB D2H): %.3f s, kernel %.3f ms" % (t1 - t0, g.kernel_ms()))
