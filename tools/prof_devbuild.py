"""rocprofv3 driver: mpt_build_and_upload of the 1 M-primitive height-field scene, three times (kernel summary of the device build)."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metalpathtracer_amd import capi
n = 501
rng = np.random.default_rng(1)
xs = np.linspace(-40, 40, n).astype(np.float32)
h = (rng.uniform(-0.4, 0.4, (n, n)) + 3.0 * np.sin(xs[:, None] * 0.2) * np.cos(xs[None, :] * 0.17)).astype(np.float32)
P = np.stack(np.broadcast_arrays(xs[None, :], h, xs[:, None]), -1)
a, b_, c, d = P[:-1, :-1], P[:-1, 1:], P[1:, :-1], P[1:, 1:]
tris = np.concatenate([np.stack([a, b_, c], 2).reshape(-1, 3, 3), np.stack([b_, d, c], 2).reshape(-1, 3, 3)])
tris = np.concatenate([tris + np.float32([0, -10, -30]), tris * np.float32(0.6) + np.float32([0, 35, -60])])
prims = np.zeros((len(tris) + 3, 3, 4), np.float32)
prims[3:, :, :3] = tris
prims[3:, 0, 3] = 1.0
prims[:3, 0, :3] = [[-15, 18, -10], [15, 18, -10], [0, 60, -20]]
prims[:3, 1, 0] = [9, 9, 10]
mats = np.zeros((len(prims), 2, 4), np.float32)
mats[:, 0, :3] = 0.7
ctx = capi.Context(0)
for k in range(3):
    print("build %d: %.2f ms device" % (k, ctx.build_and_upload(prims, mats)), flush=True)
print(ctx.accel_info())
