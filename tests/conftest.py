import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ASSETS = os.path.join(ROOT, "assets")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the oracle and the product libraries if they are missing (CPU: hipcc cross-compiles)."""
    from oracle import binding as ob
    ob.build()
    lib = os.path.join(ROOT, "metalpathtracer_amd", "lib", "libmpt_host.so")
    if not os.path.exists(lib):
        import subprocess
        subprocess.check_call(["make", "-C", ROOT, "metalpathtracer_amd/lib/libmpt_hip.so", "host"])
    yield


def scene_path(name):
    return os.path.join(ASSETS, name)


_oracle_scene_cache = {}


def oracle_scene(name):
    """Oracle-built scene (cached): (OracleScene, buffers)."""
    from oracle import binding as ob
    if name not in _oracle_scene_cache:
        sc = ob.OracleScene()
        assert sc.load_xml(scene_path(name)) == 0
        sc.build_bvh()
        _oracle_scene_cache[name] = (sc, sc.buffers())
    return _oracle_scene_cache[name]


_host_scene_cache = {}


def host_scene(name):
    """Product host-layer scene (cached): (Scene, buffers)."""
    from metalpathtracer_amd import host
    if name not in _host_scene_cache:
        sc = host.Scene()
        st, _ = host.SceneLoader.LoadSceneFromXML(scene_path(name), sc)
        assert st == 0
        sc.buildBVH()
        _host_scene_cache[name] = (sc, sc.buffers())
    return _host_scene_cache[name]


CORNELL_CAM = dict(pos=(0.0, 1.0, 3.4), fwd=(0.0, 0.0, -1.0), up=(0.0, 1.0, 0.0), vfov=40.0)


def pixel_l2(a, b):
    """SURVEY.md 8(d) parity metric: sqrt(mean over pixels of ||rgb_a - rgb_b||^2)."""
    d = a[..., :3].astype(np.float64) - b[..., :3].astype(np.float64)
    return float(np.sqrt((d * d).sum(-1).mean()))


@pytest.fixture(scope="session")
def gpu_ctx():
    from metalpathtracer_amd import capi
    ctx = capi.Context(0)  # raises loudly without a GPU: there is no CPU fallback
    yield ctx
    ctx.close()
