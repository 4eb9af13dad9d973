"""Python face of the graph-level C ABI, mirroring the reference's public surface by name:

    ONNXGraph(data:configuration:)   /root/reference/Sources/Smelter/ONNXGraph.swift:95
    ONNXGraph(contentsOf:...)        :158
    .outputShapes / .modelFormat     :69 / :58
    .metalGraph(device:)             :169   -> here `.hipGraph(device, batch)` returning an `Executor`
    graph.encode(to:sourceImages:)   README.md:43-44  -> `Executor.encode(stream, images)`
    MPSImage.toFloatArray()          Extensions/Foundation/MPSImage+Extensions.swift:9 -> `Executor.toFloatArray()`

All work happens in libsmelter_hip.so; torch only provides device tensors and the stream.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np

from . import _native as N
from ._native import SmelterError  # noqa: F401  (re-export: ONNXGraph.Errors)


@dataclass
class Shape:  # TypeDefinitions.swift:1-33
    channels: int
    width: int
    height: int
    depth: int

    def toArray(self):
        return [self.channels, self.width, self.height, self.depth]


# SM_FUSE_* bits of Configuration.fusions a caller may want to flip (smelter_hip.h has the full list)
FUSE_STAGE, FUSE_STAGE_WIDE = 512, 1024


def default_fusions() -> int:
    """SM_FUSE_DEFAULT as smelter_config_default() returns it (the SMELTER_* environment overrides applied)"""
    c = N.Config()
    N.lib().smelter_config_default(C.byref(c))
    return int(c.fusions)


@dataclass
class Configuration:  # ONNXGraph.swift:6-36 (+ executor options)
    inputConstraint: str = "none"            # "none" | "lanczos" | "bilinear" (.forceInputScale: source images of any size)
    alignCorners: bool = True                # billinearUpsamplingConfiguration.alignCorners
    dims: Dict[int, int] = field(default_factory=dict)
    fuse: bool = True
    useHipGraph: bool = True
    autotune: bool = False
    keepIntermediates: bool = False
    branchConcurrency: bool = False          # projection-shortcut convs on a second stream (measured: fork/join gaps eat the gain)
    fusions: int = -1                        # SM_FUSE_* mask (smelter_hip.h); -1 = smelter_config_default()'s (SM_FUSE_DEFAULT edited by the SMELTER_* environment overrides)
    autotuneMode: int = -1                   # 0 in situ | 1 cold; -1 = the default
    autotuneMaxCfg: int = -1                 # only the first n tile configurations; -1 = the default (all)

    def _c(self) -> N.Config:
        c = N.Config()
        N.lib().smelter_config_default(C.byref(c))
        c.input_constraint = {"none": 0, "lanczos": 1, "bilinear": 2}[self.inputConstraint]
        c.align_corners = int(self.alignCorners)
        for k, v in self.dims.items():
            c.dims[k] = v
        c.fuse, c.use_hip_graph, c.autotune = int(self.fuse), int(self.useHipGraph), int(self.autotune)
        c.keep_intermediates = int(self.keepIntermediates)
        c.branch_concurrency = int(self.branchConcurrency)
        if self.fusions >= 0:
            c.fusions = self.fusions
        if self.autotuneMode >= 0:
            c.autotune_mode = self.autotuneMode
        if self.autotuneMaxCfg >= 0:
            c.autotune_max_cfg = self.autotuneMaxCfg
        return c


class Format:  # ONNXGraph.Format (:49-52)
    onnx = 0
    mpsFlavor = 1


def _stream_ptr(stream) -> C.c_void_p:
    if stream is None:
        import torch
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)
    if hasattr(stream, "cuda_stream"):
        return C.c_void_p(stream.cuda_stream)
    return C.c_void_p(int(stream))


class ONNXGraph:
    def __init__(self, data: bytes, configuration: Optional[Configuration] = None):
        self.configuration = configuration or Configuration()
        self._h = C.c_void_p()
        L = N.lib()
        cfg = self.configuration._c()
        data = bytes(data)
        N.check(L.smelter_graph_create(data, len(data), C.byref(cfg), C.byref(self._h)))

    @classmethod
    def contentsOf(cls, path: str, configuration: Optional[Configuration] = None) -> "ONNXGraph":
        with open(path, "rb") as f:
            return cls(f.read(), configuration)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and N._lib is not None:
            N._lib.smelter_graph_destroy(h)

    @property
    def modelFormat(self) -> int:
        return N.lib().smelter_graph_format(self._h)

    @modelFormat.setter
    def modelFormat(self, fmt: int):
        N.check(N.lib().smelter_graph_set_format(self._h, fmt))

    @property
    def outputShapes(self):
        arr = (N.Shape * 8)()
        n = N.lib().smelter_graph_output_shapes(self._h, arr, 8)
        return [Shape(a.channels, a.width, a.height, a.depth) for a in arr[:min(n, 8)]]

    @property
    def numNodes(self) -> int:
        return N.lib().smelter_graph_num_nodes(self._h)

    def weightBytes(self, batch: int = 1) -> int:
        b = C.c_size_t()
        N.check(N.lib().smelter_graph_weight_bytes(self._h, batch, C.byref(b)))
        return b.value

    def packWeights(self, batch: int = 1) -> np.ndarray:
        """host image of the weight arena (BN fold + fp16 cast + OHWI + padding); needs no GPU"""
        buf = np.zeros(self.weightBytes(batch), np.uint8)
        N.check(N.lib().smelter_graph_pack_weights(self._h, batch, buf.ctypes.data_as(C.c_void_p), buf.size))
        return buf

    def defaultPlan(self, batch: int = 1) -> np.ndarray:
        """execution plan (int32 words) of a compile without autotuning; host only"""
        n = C.c_size_t()
        N.check(N.lib().smelter_graph_default_plan(self._h, batch, None, 0, C.byref(n)))
        buf = np.zeros(n.value, np.int32)
        N.check(N.lib().smelter_graph_default_plan(self._h, batch, buf.ctypes.data_as(C.POINTER(C.c_int32)), buf.size, C.byref(n)))
        return buf

    def hipGraph(self, device: int = 0, batch: int = 1, weight_arena=None, upload_weights: bool = True, plan=None) -> "Executor":
        """metalGraph(device:) — compile for `device`.  `weight_arena`: optional torch uint8 CUDA tensor of
        weightBytes(batch) bytes owned by the caller (used for the RCCL weight broadcast).  `plan`: int32 array from another
        executor's `.plan` (same graph, batch and library build): its tile-configuration picks are applied instead of autotuning."""
        ex = C.c_void_p()
        ptr = C.c_void_p(weight_arena.data_ptr()) if weight_arena is not None else None
        if plan is not None:
            p = np.ascontiguousarray(np.asarray(plan, np.int32))
            N.check(N.lib().smelter_graph_compile_with_plan(self._h, device, batch, ptr, int(upload_weights), p.ctypes.data_as(C.POINTER(C.c_int32)), p.size, C.byref(ex)))
        else:
            N.check(N.lib().smelter_graph_compile(self._h, device, batch, ptr, int(upload_weights), C.byref(ex)))
        return Executor(ex, batch, weight_arena)

    metalGraph = hipGraph  # name kept for drop-in readers


class Executor:
    def __init__(self, handle, batch, keepalive=None):
        self._h, self.batch, self._keep = handle, batch, keepalive

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and N._lib is not None:
            N._lib.smelter_exec_destroy(h)

    @property
    def outputDims(self):
        n, c, h, w = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        N.check(N.lib().smelter_exec_output_dims(self._h, C.byref(n), C.byref(c), C.byref(h), C.byref(w)))
        return n.value, c.value, h.value, w.value

    def encode(self, images, stream=None):
        """images: torch float32 CUDA tensor [batch, C, H, W] (contiguous).  Asynchronous."""
        assert images.is_cuda and images.is_contiguous() and images.dtype.is_floating_point and images.element_size() == 4
        assert images.shape[0] == self.batch
        N.check(N.lib().smelter_exec_encode_sized(self._h, _stream_ptr(stream), C.c_void_p(images.data_ptr()), images.shape[2], images.shape[3]))

    def toFloatArray(self, stream=None, layout: str = "nchw") -> np.ndarray:
        n, c, h, w = self.outputDims
        L = N.lib()
        if layout == "nchw":
            out = np.empty((n, c, h, w), np.float32)
            code = 0
        else:  # the reference's slice-packed order
            out = np.empty(L.sm_float_array_count(n, c, h, w), np.float32)
            code = 1
        N.check(L.smelter_exec_read_float(self._h, _stream_ptr(stream), out.ctypes.data_as(C.c_void_p), out.size, code))
        return out

    def readValue(self, name: str, stream=None) -> np.ndarray:
        n = self.batch
        L = N.lib()
        c, h, w = C.c_int32(), C.c_int32(), C.c_int32()
        cap = 1 << 20
        while True:
            buf = np.empty(cap, np.float32)
            rc = L.smelter_exec_read_value(self._h, _stream_ptr(stream), name.encode(), buf.ctypes.data_as(C.c_void_p), cap,
                                           C.byref(c), C.byref(h), C.byref(w))
            need = n * max(c.value, 1) * max(h.value, 1) * max(w.value, 1)
            if rc == -13 and need > cap:
                cap = need
                continue
            N.check(rc)
            return buf[:need].reshape(n, c.value, h.value, w.value).copy()

    # introspection
    @property
    def plan(self) -> np.ndarray:
        """the tile configuration of every step (autotuner's picks included) as int32 words: feed it to ONNXGraph.hipGraph(plan=...) on another rank"""
        n = C.c_size_t()
        N.check(N.lib().smelter_exec_get_plan(self._h, None, 0, C.byref(n)))
        buf = np.zeros(n.value, np.int32)
        N.check(N.lib().smelter_exec_get_plan(self._h, buf.ctypes.data_as(C.POINTER(C.c_int32)), buf.size, C.byref(n)))
        return buf

    numLaunches = property(lambda s: N.lib().smelter_exec_num_launches(s._h))
    numOps = property(lambda s: N.lib().smelter_exec_num_ops(s._h))
    weightBytes = property(lambda s: N.lib().smelter_exec_weight_bytes(s._h))
    weightArenaPtr = property(lambda s: N.lib().smelter_exec_weight_arena(s._h))
    activationBytes = property(lambda s: N.lib().smelter_exec_activation_bytes(s._h))
    flops = property(lambda s: N.lib().smelter_exec_flops(s._h))
    algorithmicBytes = property(lambda s: N.lib().smelter_exec_algorithmic_bytes(s._h))
    launchBytes = property(lambda s: N.lib().smelter_exec_launch_bytes(s._h))

    @property
    def valueNames(self):
        """tensors that exist as fp16 after fusion: graph input, every kernel output, and the hidden tensors a fused launch rounds in registers / LDS
        without storing them — the list a parity oracle rounds to emulate the executor"""
        L = N.lib()
        return [L.smelter_exec_value_name(self._h, i).decode() for i in range(L.smelter_exec_num_values(self._h))]

    @property
    def storedValueNames(self):
        """the subset of valueNames that has storage in the activation arena (readValue works on exactly these)"""
        L = N.lib()
        return [L.smelter_exec_value_name(self._h, i).decode() for i in range(L.smelter_exec_num_values(self._h)) if L.smelter_exec_value_stored(self._h, i)]

    @property
    def persistentRuns(self):
        """the persistent runs of this executor (executor pass 3k) with what the compile's A/B measured: a list of dicts, one per run (smelter_exec_run_info)"""
        L = N.lib()
        out = []
        for i in range(L.smelter_exec_num_runs(self._h)):
            r = N.RunInfo()
            N.check(L.smelter_exec_run_info(self._h, i, C.byref(r)))
            why = [name for bit, name in ((1, "slower"), (2, "timeout"), (4, "claimed_mode_only"), (8, "device"), (16, "imported_plan")) if r.why_off & bit]
            out.append({"first_step": r.first_step, "layers": r.layers, "on": bool(r.on), "timed": bool(r.timed), "samples": r.samples,
                        "t_run_us": round(r.t_run_us, 1), "t_layers_us": round(r.t_layers_us, 1), "mode_bits": r.mode_bits, "first_err": r.err_bits, "why_off": why})
        return out

    def opDesc(self, idx: int) -> str:
        return N.lib().smelter_exec_op_desc(self._h, idx).decode()

    def profile(self, images, iters: int = 10, stream=None):
        """in-situ per-launch times (ms): [input conversion, op 0, op 1, ...], HIP events between launches"""
        n = self.numOps + 1
        arr = (C.c_float * n)()
        N.check(N.lib().smelter_exec_profile(self._h, _stream_ptr(stream), C.c_void_p(images.data_ptr()), iters, arr, n))
        return list(arr)

    def timeOp(self, idx: int, iters: int = 10, stream=None) -> float:
        ms = C.c_float()
        N.check(N.lib().smelter_exec_time_op(self._h, _stream_ptr(stream), idx, iters, C.byref(ms)))
        return ms.value
