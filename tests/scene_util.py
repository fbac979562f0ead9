"""Shared helpers: seeded scenes as numpy planes for the oracle / hostsim / HIP comparisons."""
import functools

import numpy as np


@functools.lru_cache(maxsize=8)
def cpu_scene(W, H, shadow_dim=512, cube_dim=64):
    from crychic_renderer_amd import scene
    return scene.make_scene(W, H, shadow_dim=shadow_dim, cube_dim=cube_dim, device="cpu")


def np_planes(planes):
    """torch planes -> numpy views in the dtypes the oracle expects."""
    out = {}
    for k in ("depth", "shadow"):
        out[k] = planes[k].cpu().numpy().view(np.uint32)
    out["normal"] = planes["normal"].cpu().numpy()
    for k in ("g0", "g1", "g2", "cube", "randvec"):
        out[k] = planes[k].cpu().numpy()
    return out
