"""MI355X-native ray-trace path behind the reference renderer's per-frame call.

Python only wraps the C-ABI of librt_gfx950.so (include/rt_capi.h) and the C++
host API's harness; all rendering happens in the HIP kernels.
"""
from ._native import (RT_COUNTER_CNT, RT_COUNTER_NAMES, RtCamera, RtDeviceOpts, RtFrameAux, RtLight, RtSceneDesc,  # noqa: F401
                      load)
from .api import Context, Experiment, HostScene, MultiContext, RtError, pack_scene, tiles_of_rank  # noqa: F401
