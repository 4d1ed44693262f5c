"""ctypes binding of libptts.so (C ABI: include/ptts.h).  Fails loudly when the HIP library is
missing: there is no CPU fallback for the hot path."""

from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
LIB_PATH = PKG / "libptts.so"
SRC = PKG / "csrc" / "ptts.hip"
HEADER = PKG.parent / "include" / "ptts.h"

# -ffp-contract=on: FMA contraction is decided per source expression, identically for every unrolled instance.
# hipcc's default ("fast") contracts opportunistically in the backend, and e.g. the LayerNorm statistics of the two
# row tiles of one wave then round differently: a sequence's output depended (at the 1e-7 level) on which row tile of
# a workgroup it landed in (tests/test_gpu_parity.py::test_full_size_batch64_properties).
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-Wno-unused-value", "-Wno-unused-result",
               "-shared", "-fPIC"]


class PttsConfig(C.Structure):
    _fields_ = [
        ("d_model", C.c_int32), ("num_heads", C.c_int32), ("num_layers", C.c_int32), ("ff_dim", C.c_int32),
        ("ldim", C.c_int32), ("flow_dim", C.c_int32), ("flow_depth", C.c_int32), ("max_period", C.c_float),
        ("m_dim", C.c_int32), ("m_heads", C.c_int32), ("m_layers", C.c_int32), ("m_ff", C.c_int32),
        ("m_context", C.c_int32), ("m_max_period", C.c_float),
        ("n_filters", C.c_int32), ("ratios", C.c_int32 * 3), ("kernel_size", C.c_int32),
        ("res_kernel_size", C.c_int32), ("last_kernel_size", C.c_int32), ("compress", C.c_int32),
        ("upsample_stride", C.c_int32),
    ]


class PttsTensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("d_data", C.c_void_p), ("numel", C.c_int64)]


# every symbol include/ptts.h declares: (restype, argtypes)
_P = C.c_void_p
PROTOTYPES = {
    "ptts_abi_version": (C.c_int, []),
    "ptts_last_error": (C.c_char_p, []),
    "ptts_create": (C.c_int, [C.POINTER(PttsConfig), C.POINTER(PttsTensor), C.c_int32, C.c_int32, C.POINTER(_P)]),
    "ptts_create_ex": (C.c_int, [C.POINTER(PttsConfig), C.POINTER(PttsTensor), C.c_int32, C.c_int32, C.c_int32, C.POINTER(_P)]),
    "ptts_engine_save": (C.c_int, [_P, C.c_char_p]),
    "ptts_create_from_file": (C.c_int, [C.c_char_p, C.c_int32, C.POINTER(_P)]),
    "ptts_destroy": (None, [_P]),
    "ptts_lm_state_create": (C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(_P)]),
    "ptts_lm_state_destroy": (None, [_P]),
    "ptts_lm_state_reset": (C.c_int, [_P, _P]),
    "ptts_lm_state_import": (C.c_int, [_P, C.c_int32, _P, C.c_int32, C.c_int32, _P]),
    "ptts_lm_state_export": (C.c_int, [_P, C.c_int32, _P, C.c_int32, _P]),
    "ptts_lm_state_copy": (C.c_int, [_P, _P, _P]),
    "ptts_lm_state_copy_row": (C.c_int, [_P, C.c_int32, _P, _P]),
    "ptts_lm_state_copy_row_from": (C.c_int, [_P, C.c_int32, _P, C.c_int32, _P]),
    "ptts_lm_state_offsets": (C.c_int, [_P, C.POINTER(C.c_int32), _P]),
    "ptts_lm_prefill": (C.c_int, [_P, _P, _P, C.c_int32, _P]),
    "ptts_lm_decode_step": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_float, _P, _P, _P, _P]),
    "ptts_lm_latent_ptr": (_P, [_P]),
    "ptts_lm_set_noise": (C.c_int, [_P, C.c_float, C.c_uint64]),
    "ptts_profile_start": (C.c_int, [_P]),
    "ptts_profile_stop": (C.c_int64, [_P, C.c_char_p, C.c_int64]),
    "ptts_mimi_state_create": (C.c_int, [_P, C.c_int32, C.POINTER(_P)]),
    "ptts_mimi_state_destroy": (None, [_P]),
    "ptts_mimi_state_reset": (C.c_int, [_P, _P]),
    "ptts_mimi_decode": (C.c_int, [_P, _P, _P, _P, _P]),
    "ptts_encode_voice": (C.c_int, [_P, _P, C.c_int64, _P, _P, C.POINTER(C.c_int32), _P]),
    "ptts_graph_capture_lm_step": (C.c_int, [_P, _P, _P, C.c_int32, C.c_float, _P, _P, _P, C.POINTER(_P)]),
    "ptts_graph_capture_mimi": (C.c_int, [_P, _P, _P, _P, C.POINTER(_P)]),
    "ptts_graph_capture_pipelined": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_float, _P, _P, _P, _P, _P, C.POINTER(_P)]),
    "ptts_graph_launch": (C.c_int, [_P, _P]),
    "ptts_graph_destroy": (None, [_P]),
    "ptts_mimi_state_reset_row": (C.c_int, [_P, C.c_int32, _P]),
    "ptts_lm_state_set_row_active": (C.c_int, [_P, C.c_int32, C.c_int32, _P]),
    "ptts_mimi_set_pcm_i16": (C.c_int, [_P, _P]),
    "ptts_tune": (C.c_int, [_P, C.c_int32, _P]),
    "ptts_tune_streams": (C.c_int, [_P, C.c_int32, _P, _P]),
    "ptts_streams_overlap": (C.c_int, [_P, _P, _P]),
    "ptts_tune_prefill": (C.c_int, [_P, C.c_int32, C.c_int32, _P]),
    "ptts_tune_log": (C.c_char_p, [_P]),
    "ptts_tune_clear": (None, [_P]),
    "ptts_tune_export": (C.c_int64, [_P, C.c_char_p, C.c_int64]),
    "ptts_tune_import": (C.c_int, [_P, C.c_char_p]),
    "ptts_tune_version": (C.c_int, []),
    "ptts_set_option": (C.c_int, [_P, C.c_char_p, C.c_int32]),
    "ptts_lm_state_error": (C.c_int, [_P, _P]),
    "ptts_debug_set_error": (C.c_int, [_P, C.c_int32, _P]),
    "ptts_stream_create_masked": (C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(_P)]),
    "ptts_stream_destroy": (C.c_int, [_P]),
    "ptts_sync": (C.c_int, [_P, _P]),
    "ptts_engine_stream": (_P, [_P]),
    "ptts_copy_to_host_async": (C.c_int, [_P, _P, _P, C.c_int64, _P]),
    "ptts_embed_tokens": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int64, _P, _P]),
    "ptts_timer_start": (C.c_int, [_P, _P]),
    "ptts_timer_stop_ms": (C.c_int, [_P, _P, C.POINTER(C.c_float)]),
    "ptts_debug_read": (C.c_int64, [_P, _P, C.c_int32, C.c_char_p, _P, C.c_int64, C.POINTER(C.c_int32),
                                    C.POINTER(C.c_int32), _P]),
    "ptts_lm_weight_bytes": (C.c_int64, [_P]),
    "ptts_mimi_weight_bytes": (C.c_int64, [_P]),
}


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile csrc/*.hip for gfx950 in-tree (hipcc cross-compiles without a GPU).  Every .hip file is one translation
    unit (ptts.hip = host side + the round-1/2 kernels; newer kernel families live in their own files behind plain C++
    launcher functions declared in ptts_ext.h), compiled in parallel to csrc/.obj/*.o and linked into libptts.so; a
    unit is recompiled when it or any header is newer than its object."""
    from concurrent.futures import ThreadPoolExecutor

    srcs = sorted(SRC.parent.glob("*.hip"))
    hdrs = [*sorted(SRC.parent.glob("*.h")), HEADER]
    obj_dir = SRC.parent / ".obj"
    obj_dir.mkdir(exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cflags = [f for f in HIPCC_FLAGS if f != "-shared"]
    newest_hdr = max(h.stat().st_mtime for h in hdrs)

    def stale(src):
        o = obj_dir / (src.stem + ".o")
        return force or not o.exists() or o.stat().st_mtime < max(src.stat().st_mtime, newest_hdr)

    todo = [s for s in srcs if stale(s)]
    objs = [obj_dir / (s.stem + ".o") for s in srcs]
    if not todo and LIB_PATH.exists() and all(LIB_PATH.stat().st_mtime >= o.stat().st_mtime for o in objs):
        return LIB_PATH

    def compile_one(src):
        cmd = [hipcc, *cflags, "-c", "-o", str(obj_dir / (src.stem + ".o")), str(src)]
        if verbose:
            print(" ".join(cmd), flush=True)
        return src, subprocess.run(cmd, capture_output=True, text=True)

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(todo)))) as ex:
        for src, r in ex.map(compile_one, todo):
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed on {src.name}:\n{r.stdout}\n{r.stderr}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB_PATH), *map(str, objs)]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc link failed:\n{r.stdout}\n{r.stderr}")
    return LIB_PATH


_lib = None


def load() -> C.CDLL:
    """Loads libptts.so; raises if it is absent (the product has no other path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  The MI355X hot path has no CPU fallback."
        )
    # torch ships its own HIP runtime: import it first so that libptts binds to the SAME libamdhip64
    # (two runtimes in one process do not share devices, streams or allocations)
    import torch  # noqa: F401

    # PTTS_LIB_PATH: load another build of the same sources (A/B runs of compile-time experiments)
    lib = C.CDLL(os.environ.get("PTTS_LIB_PATH") or str(LIB_PATH))
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.ptts_abi_version() != 1:
        raise RuntimeError("libptts ABI version mismatch")
    _lib = lib
    return lib


class PttsError(RuntimeError):
    pass


def check(rc: int):
    if rc < 0:
        msg = load().ptts_last_error().decode()
        if rc == -5:
            raise ValueError(msg)  # capacity errors mirror the reference's ValueError family
        raise PttsError(f"libptts error {rc}: {msg}")
    return rc
