"""ctypes binding of ``libmhmr.so`` (C ABI declared in include/mhmr.h).

The HIP library is the product: there is NO CPU / PyTorch fallback.  If the shared object is missing or an
entry point fails, loading raises and every op raises -- loudly, by design.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libmhmr.so")
SOURCES = ["gemm.hip", "gemm256.hip", "attention.hip", "attention_f32.hip", "vit_misc.hip", "vit_cls.hip", "hph.hip", "lbs.hip", "preprocess.hip", "evalm.hip", "anny.hip", "capi.hip"]
HEADERS = ["mhmr_common.h", "mhmr_internal.h", "ln_stats.h", os.path.join("..", "..", "include", "mhmr.h")]

VERSION = 106                       # include/mhmr.h MHMR_VERSION (struct layouts and entry-point semantics)
DT_BF16, DT_F16 = 0, 1
ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
EPI_OP16, EPI_OP16_GELU, EPI_OP16_RELU, EPI_RESID, EPI_PATCH, EPI_F32, EPI_VT, EPI_OP16_QK = range(8)
ATTN_QSCALE = 0.18033688011112042   # include/mhmr.h MHMR_ATTN_QSCALE

_vp, _fp, _ip, _i, _f = C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float  # all device pointers are void*


class MhmrError(RuntimeError):
    pass


#: No SLP vectoriser in any translation unit (csrc/mhmr_common.h has the reasons and refuses a build without -DMHMR_NO_SLP): its
#: op_sel-swizzled v_pk_*_f32 code returned wrong values in two kernels on gfx950, and the scalar build is 2 % faster.
COMMON_FLAGS = ["-fno-slp-vectorize", "-DMHMR_NO_SLP"]
#: per-translation-unit extra flags
EXTRA_FLAGS = {}


def source_hash() -> str:
    """sha256 prefix over the contents of every source / header of the library and the compile flags: the identity of a BUILD RECIPE,
    the same on every machine and in every checkout path (unlike the .so's own hash, which embeds path-derived symbol ids).  It is
    compiled into the library (``mhmr_source_hash()``), so a shipped ``libmhmr.so`` says which sources it was made from."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(SOURCES) + sorted(HEADERS):
        with open(os.path.normpath(os.path.join(CSRC, name)), "rb") as f:
            h.update(os.path.basename(name).encode() + b"\0" + f.read() + b"\0")
    h.update(repr((COMMON_FLAGS, sorted(EXTRA_FLAGS.items()))).encode())
    return h.hexdigest()[:16]


def built_source_hash(path: str | None = None) -> str | None:
    """The source hash compiled into an existing libmhmr.so (None: no library, or one from before the hash existed).  Read from the
    file's bytes (capi.hip plants the marker string), NOT through dlopen: a library loaded here would be the one the process keeps
    even after build() has replaced the file."""
    import re
    path = path or LIB_PATH
    if not os.path.isfile(path):
        return None
    with open(path, "rb") as f:
        m = re.search(rb"MHMR_SOURCE_HASH=([0-9a-f]{16})", f.read())
    return m.group(1).decode() if m else None


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP translation unit for gfx950 (hipcc cross-compiles without a GPU; one object per source, in parallel) and link
    csrc/libmhmr.so.  An existing library is kept only if the source hash compiled into it equals the hash of the sources beside it
    (file times say nothing about a library that travelled from another machine)."""
    from concurrent.futures import ThreadPoolExecutor
    want = source_hash()
    if not force and built_source_hash() == want:
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f'-DMHMR_SOURCE_HASH="{want}"'] + COMMON_FLAGS

    def compile_one(name):
        obj = os.path.join(objdir, name.replace(".hip", ".o"))
        cmd = base + EXTRA_FLAGS.get(name, []) + ["-c", os.path.join(CSRC, name), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise MhmrError(f"hipcc failed on {name}:\n" + res.stdout + res.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise MhmrError("hipcc link failed:\n" + res.stdout + res.stderr)
    return LIB_PATH


class VitBlock(C.Structure):
    _fields_ = ([(n, _vp) for n in ("ln1_w", "ln1_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "ls1", "ln2_w", "ln2_b",
                                    "fc1_w", "fc1_b", "fc2_w", "fc2_b", "ls2", "v_w2", "proj_w2")] +
                [("flags", _i), ("qkv_colsum", _vp), ("fc1_colsum", _vp), ("v_w8", _vp), ("proj_w8", _vp), ("v_w8_scale", _i), ("proj_w8_scale", _i)])


class VitDesc(C.Structure):
    _fields_ = ([(n, _i) for n in ("dtype", "B", "S", "C", "H", "L", "G", "N", "T", "Tp", "Kp")] +
                [("patch_w", _vp), ("patch_b", _vp), ("cls_pos0", _vp), ("pos", _vp), ("blocks", C.POINTER(VitBlock)),
                 ("norm_w", _vp), ("norm_b", _vp)] +
                [(n, _vp) for n in ("a_patch", "resid", "xn", "qk", "vt", "att", "hid", "attn_flags", "pstats", "rowstats")] +
                [("lo8", _i), ("x3", _i), ("qkv32", _vp), ("hid32", _vp), ("splitk", _vp), ("splitk_bytes", C.c_longlong), ("cpad", _i), ("v16", _vp), ("cls_pstats", _vp)])


class HphLayer(C.Structure):
    _fields_ = [(n, _vp) for n in ("ln_sa_w", "ln_sa_b", "to_qkv", "sa_out_w", "sa_out_b", "ln_ca_w", "ln_ca_b", "to_kv16",
                                   "to_q", "ca_out_w", "ca_out_b", "ln_ff_w", "ln_ff_b", "ff1_w", "ff1_b", "ff2_w", "ff2_b")]


class HphDesc(C.Structure):
    _fields_ = ([(n, _i) for n in ("dtype", "C", "G", "N", "Kc", "dim", "heads", "mlp", "depth", "nb", "Ktok", "Ndec", "patch",
                                   "nearness")] + [("fn", _f)] +
                [(n, _vp) for n in ("off1_w", "off1_b", "off2_w", "off2_b", "cq_x", "cq_y", "cv_x", "cv_y", "init_tail",
                                    "tok_w", "tok_b")] + [("layers", C.POINTER(HphLayer))] +
                [(n, _vp) for n in ("dec_w", "dec_b", "zc", "token", "x", "xn", "t1", "t2", "kv", "dec", "det_row", "nvalid")] +
                [("cam_dim", _i)])


class LbsConsts(C.Structure):
    _fields_ = ([(n, _i) for n in ("V", "Vp", "Vl", "Kb", "nb", "Kinf", "center_joint")] +
                [(n, _vp) for n in ("basis16", "vtemp", "J0", "JS", "parents", "skin_idx", "skin_w", "skin16", "xbary", "pose_tasks")] +
                [("pose_levels", _i)])


_SIGS = {
    "mhmr_version": ([], _i),
    "mhmr_source_hash": ([], C.c_char_p),
    "mhmr_vit_forward": ([C.POINTER(VitDesc), _vp, _vp, _vp, _i, _vp], _i),
    "mhmr_gemm16": ([_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp], _i),
    "mhmr_gemm16_ex": ([_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp], _i),
    "mhmr_gemm16_ln": ([_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp], _i),
    "mhmr_gemm16_lo8": ([_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp], _i),
    "mhmr_gemm16_masked": ([_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp], _i),
    "mhmr_qkv16": ([_vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp], _i),
    "mhmr_splitk_workspace_bytes": ([_i, _i, _i], C.c_longlong),
    "mhmr_gemm16_splitk_resid": ([_vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _f, _vp, C.c_longlong, _i, _vp], _i),
    "mhmr_ln_stats": ([_vp, _vp, _vp, _i, _i, _i, _i, _f, _vp], _i),
    "mhmr_cls_linear16": ([_vp, C.c_longlong, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, C.c_longlong, _i, _i, _vp, _i, _i, _i, _i, _i, _vp], _i),
    "mhmr_attention16": ([_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp], _i),
    "mhmr_attention16_ex": ([_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _vp], _i),
    "mhmr_attention_flag_count": ([_i, _i, _i], _i),
    "mhmr_attention16_pitch": ([_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _vp], _i),
    "mhmr_layernorm16_pitch": ([_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp], _i),
    "mhmr_attention_f32": ([_vp, _vp, _i, _i, _i, _i, _i, _i, _vp], _i),
    "mhmr_layernorm16_pair": ([_vp, _vp, _vp, _vp, _i, _i, _f, _i, _vp], _i),
    "mhmr_gelu16_pair": ([_vp, _vp, C.c_longlong, _i, _i, _vp], _i),
    "mhmr_layernorm16": ([_vp, _vp, _vp, _vp, _i, _i, _f, _i, _vp], _i),
    "mhmr_detect_scores": ([_vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp], _i),
    "mhmr_detect_count": ([_vp, _i, _i, _i, _f, _vp, _vp], _i),
    "mhmr_detect_write": ([_vp, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp], _i),
    "mhmr_detect_write_cap": ([_vp, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _i, _vp], _i),
    "mhmr_person_groups": ([_vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _vp], _i),
    "mhmr_camera_embed": ([_vp, _vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _vp], _i),
    "mhmr_hph_forward": ([C.POINTER(HphDesc), _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _vp, _i, _vp, _i,
                          _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp], _i),
    "mhmr_xattn_layers_forward": ([C.POINTER(HphLayer)] + [_i] * 8 + [_vp] * 7 + [_i, _i, _vp, _i, _i, _vp], _i),
    "mhmr_linear_f32": ([_vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp], _i),
    "mhmr_layernorm_f32": ([_vp, _vp, _vp, _vp, _i, _i, _f, _vp], _i),
    "mhmr_lbs_forward": ([C.POINTER(LbsConsts)] + [_vp] * 7 + [_i] + [_vp] * 8 + [_vp], _i),
    "mhmr_lbs_forward_fused": ([C.POINTER(LbsConsts)] + [_vp] * 7 + [_i] + [_vp] * 8 + [_vp, _vp], _i),
    "mhmr_preprocess_u8": ([_vp, _i, _i, _vp, _vp, _i, _vp, _vp, _i] + [_i] * 7 + [_vp, _vp, _vp, _vp], _i),
    "mhmr_eval_mesh_errors": ([_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp], _i),
    "mhmr_anny_scores": ([_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp], _i),
    "mhmr_anny_camera": ([_vp, _i, _i, _f, _vp, _vp, _vp], _i),
    "mhmr_anny_decode": ([_vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i] + [_vp] * 6 + [_vp], _i),
    "mhmr_prof_enable": ([_i], _i),
    "mhmr_prof_collect": ([C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)], _i),
}

#: every symbol include/mhmr.h declares
EXPORTS = tuple(_SIGS.keys())

_lib = None


def lib() -> C.CDLL:
    """Load (once) the in-tree libmhmr.so; raise if it is absent -- there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise MhmrError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(hipcc --offload-arch=gfx950). The Multi-HMR path has no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (args, res) in _SIGS.items():
            fn = getattr(l, name)          # AttributeError if the symbol is not exported
            fn.argtypes, fn.restype = args, res
        if l.mhmr_version() != VERSION:
            raise MhmrError(f"{LIB_PATH} is version {l.mhmr_version()}, this package binds version {VERSION} (include/mhmr.h): rebuild")
        _lib = l
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        kind = {-1: "bad argument", -2: "bad shape"}.get(rc, f"hipError_t {rc}")
        raise MhmrError(f"{what} failed: {kind}")


def ptr(t) -> int | None:
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()
