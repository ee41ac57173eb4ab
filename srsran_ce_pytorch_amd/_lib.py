"""ctypes binding of libce_hip.so (C ABI: include/ce_hip.h) and the in-tree build recipe.

The product path has no CPU fallback: if the library is missing or fails to load,
``load()`` raises and every estimator entry point with it.
"""
from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
INCLUDE = PKG_DIR.parent / "include"
LIB_PATH = Path(os.environ.get("CE_HIP_LIB", CSRC / "libce_hip.so"))   # override: diagnostic builds (tools/)
# the same library with the tuning / A-B knobs of include/ce_hip.h compiled in (-DCE_TUNING_KNOBS: environment variables
# read at plan creation).  Diagnostic only -- tests/test_hip_tiers.py and the dev tools load it through CE_HIP_LIB; the
# shipped libce_hip.so never reads the environment.
KNOBS_LIB_PATH = CSRC / "libce_hip_knobs.so"
# the estimation kernel template (ce_estimate_kernel.h) is instantiated in slices, one translation unit each, so the
# units compile concurrently (ce_inst.inc)
SOURCES = ["ce_api.hip", "ce_denoise.hip", "ce_inst_reg_h1_f0.hip", "ce_inst_reg_h1_f1.hip", "ce_inst_reg_h1_f1w.hip", "ce_inst_reg_h2_f0.hip",
           "ce_inst_reg_h2_f1.hip", "ce_inst_gen_h1.hip", "ce_inst_gen_h2.hip", "ce_inst_narrow.hip"]
HEADERS = ["ce_plan.h", "ce_estimate_kernel.h", "ce_narrow_kernel.h", "ce_inst.inc"]

CE_ABI_VERSION = 2
CE_MAX_CDM, CE_MAX_HOPS, CE_MAX_SYMBOLS = 2, 2, 14
SMOOTHING = {"none": 0, "mean": 1, "filter": 2, "mmse": 3}   # "mmse": extension, not in the reference
INTERP = {"linear": 0, "cnn": 1}
CE_ERR_INVALID, CE_ERR_UNSUPPORTED, CE_ERR_HIP, CE_ERR_NOMEM = -1, -2, -3, -4


class HopDesc(C.Structure):
    _fields_ = [("dmrs_symbols", C.c_uint8 * CE_MAX_SYMBOLS), ("re_mask", C.c_uint16 * CE_MAX_CDM),
                ("prb_start", C.c_int32), ("n_prbs", C.c_int32), ("mask_prbs", C.POINTER(C.c_uint8)),
                ("start_symbol", C.c_int32), ("n_alloc_symbols", C.c_int32)]


class PlanDesc(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("device", C.c_int32), ("n_prb_grid", C.c_int32), ("n_sym", C.c_int32),
                ("n_layers", C.c_int32), ("n_hops", C.c_int32), ("smoothing", C.c_int32),
                ("cfo_compensate", C.c_int32), ("interp", C.c_int32), ("reserved0", C.c_int32),
                ("scs_hz", C.c_double), ("beta_dmrs", C.c_double), ("cp_ms", C.c_double * CE_MAX_SYMBOLS),
                ("cnn_smoothing_alpha", C.c_double), ("mmse_delay_spread_s", C.c_double),
                ("mmse_noise_to_signal", C.c_double), ("hop", HopDesc * CE_MAX_HOPS)]


class PlanInfo(C.Structure):
    _fields_ = [("n_sc", C.c_int32), ("n_re", C.c_int32), ("n_dmrs_total", C.c_int32), ("cfo_estimated", C.c_int32),
                ("lds_bytes", C.c_int32), ("threads", C.c_int32), ("alg_bytes_per_item", C.c_int64),
                ("pilot_bytes_per_slot", C.c_int64)]


class PlanHostView(C.Structure):
    _fields_ = [("n_re", C.c_int32), ("n_dmrs_total", C.c_int32), ("n_pils", C.c_int32), ("rc_len", C.c_int32),
                ("reg_nd", C.c_int32), ("lds_bytes", C.c_int32), ("scratch_bytes", C.c_int32), ("filt_windowed", C.c_int32),
                ("cfo_estimated", C.c_int32), ("narrow", C.c_int32),
                ("ta_nres", C.c_int32 * CE_MAX_HOPS), ("contig", C.c_int32 * CE_MAX_HOPS),
                ("last_idx", (C.c_int32 * CE_MAX_CDM) * CE_MAX_HOPS),
                ("r_ord", ((C.c_int32 * 12) * CE_MAX_CDM) * CE_MAX_HOPS),
                ("alpha", ((C.c_float * 12) * CE_MAX_CDM) * CE_MAX_HOPS),
                ("rc", C.c_double * 31), ("sst", C.c_double * CE_MAX_SYMBOLS), ("two_pi_nsamples", C.c_double * CE_MAX_HOPS),
                ("n_pilots", C.c_double), ("noise_den", C.c_double), ("mmse_w", ((C.c_float * 32) * 32) * 2)]


EXPORTS = ["ce_plan_create", "ce_plan_destroy", "ce_plan_get_info", "ce_plan_derive_host", "ce_estimate_batch",
           "ce_estimate_batch_stages", "ce_time_batch", "ce_last_error", "ce_abi_version"]
EXPORTS_DENOISE = ["ce_denoiser_create", "ce_denoiser_destroy", "ce_denoise_batch"]   # include/ce_denoise.h (extension)


# per-source compiler flags: the denoiser's MFMA results feed vector instructions straight away, so keep them in
# VGPRs (the default AGPR form costs four v_accvgpr_read per 16x16 tile in a kernel bound by vector-instruction issue)
EXTRA_FLAGS = {"ce_denoise.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
               # wide single-hop FIR kernels (the headline's among them): max-ILP scheduling, 2-3 % faster in process; the narrow tiers lose with it
               "ce_inst_reg_h1_f1w.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
               # the wave-per-item kernel is bound by vector-instruction issue: the SLP vectoriser's packed-f32 forms cost a register
               # move or two per packed operation there (850 v_mov in one kernel against 465 without) -- in process 10-18 % faster on
               # the one-layer two-hop shapes without it, 0-4 % on the multi-layer ones (profiles/round3_narrow_kernel_ab.txt)
               "ce_inst_narrow.hip": ["-fno-slp-vectorize"],
               # the same for the two-hop and the re-read units (in process, all units without it: two-hop wide tiers -1...-4 %, 2 hops x 40
               # PRB -7 %, iterated in-painting -5 %, 4 layers x 2 hops -2 %; the one-hop register units are indifferent and keep the default;
               # profiles/round3_noslp_ab.txt)
               "ce_inst_reg_h2_f0.hip": ["-fno-slp-vectorize"], "ce_inst_reg_h2_f1.hip": ["-fno-slp-vectorize"],
               "ce_inst_gen_h2.hip": ["-fno-slp-vectorize"],
               # the one-hop re-read unit also without the loop vectoriser (its staged writer's store loop otherwise pairs iterations into
               # v_pk_mul/fma_f32 at ~50 register moves per 8 stores): 2 / 4 layers at full band -1 ... -2 % in two in-process A/Bs, bit-identical;
               # the two-hop unit measured +2 % with it and keeps the vectoriser (profiles/round3_reread_batched_loads_ab.txt)
               "ce_inst_gen_h1.hip": ["-fno-slp-vectorize", "-fno-vectorize"]}


def build(force: bool = False, verbose: bool = False, extra_flags=(), out: Path | None = None) -> Path:
    """hipcc --offload-arch=gfx950 -> csrc/libce_hip.so (cross-compiles without a GPU): one object per source
    (compiled concurrently, each with its own flags), then one link."""
    from concurrent.futures import ThreadPoolExecutor

    srcs = [CSRC / s for s in SOURCES]
    deps = srcs + [CSRC / h for h in HEADERS] + [INCLUDE / "ce_hip.h", INCLUDE / "ce_denoise.h", Path(__file__)]
    shipped = CSRC / "libce_hip.so"                         # build() always writes the in-tree library, whatever CE_HIP_LIB selects for loading
    target = Path(out) if out is not None else shipped
    if out is None and not force and all(t.exists() and all(t.stat().st_mtime >= d.stat().st_mtime for d in deps) for t in (shipped, KNOBS_LIB_PATH)):
        return shipped
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    objdir = CSRC / ".build" if out is None else Path(str(target) + ".obj")   # diagnostic builds (tools/) keep their own objects
    objdir.mkdir(exist_ok=True)
    common = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{INCLUDE}", f"-I{CSRC}", *extra_flags]

    def compile_one(src: Path) -> Path:
        obj = objdir / (src.stem + ".o")
        cmd = common + EXTRA_FLAGS.get(src.name, []) + ["-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
        return obj

    def compile_knobs() -> Path:                               # ce_api.hip is the only unit that looks at the knobs
        obj = objdir / "ce_api_knobs.o"
        cmd = common + ["-DCE_TUNING_KNOBS=1", "-c", str(CSRC / "ce_api.hip"), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(srcs) + 1, os.cpu_count() or 4)) as pool:
        knobs_obj = pool.submit(compile_knobs) if out is None else None
        objs = list(pool.map(compile_one, srcs))
        knobs_obj = knobs_obj.result() if knobs_obj is not None else None

    def link(objects, dst):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(dst)] + [str(o) for o in objects]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)

    link(objs, target)
    if knobs_obj is not None:
        link([knobs_obj if o.name == "ce_api.o" else o for o in objs], KNOBS_LIB_PATH)
    return target


_lib = None


def load() -> C.CDLL:
    """Load the HIP library; raise loudly when it is absent (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(f"{LIB_PATH} is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(the estimator has no CPU fallback)")
    lib = C.CDLL(str(LIB_PATH))
    vp, i64p, dp = C.c_void_p, C.POINTER(C.c_int64), C.c_void_p
    lib.ce_plan_create.argtypes = [C.POINTER(PlanDesc), C.POINTER(vp)]
    lib.ce_plan_create.restype = C.c_int
    lib.ce_plan_destroy.argtypes = [vp]
    lib.ce_plan_destroy.restype = None
    lib.ce_plan_get_info.argtypes = [vp, C.POINTER(PlanInfo)]
    lib.ce_plan_get_info.restype = C.c_int
    lib.ce_plan_derive_host.argtypes = [C.POINTER(PlanDesc), C.POINTER(PlanHostView)]
    lib.ce_plan_derive_host.restype = C.c_int
    batch = [vp, vp, i64p, vp, i64p, C.c_int64, C.c_int32, vp, dp, dp, dp, dp, dp, vp]
    lib.ce_estimate_batch.argtypes = batch
    lib.ce_estimate_batch.restype = C.c_int
    lib.ce_estimate_batch_stages.argtypes = batch[:-1] + [vp, dp, vp]
    lib.ce_estimate_batch_stages.restype = C.c_int
    lib.ce_time_batch.argtypes = batch + [C.c_int32, C.c_int32, C.POINTER(C.c_double)]
    lib.ce_time_batch.restype = C.c_int
    lib.ce_last_error.argtypes = []
    lib.ce_last_error.restype = C.c_char_p
    lib.ce_abi_version.argtypes = []
    lib.ce_abi_version.restype = C.c_int
    fp = C.POINTER(C.c_float)
    lib.ce_denoiser_create.argtypes = [C.c_int32, fp, fp, fp, fp, fp, fp, C.POINTER(vp)]
    lib.ce_denoiser_create.restype = C.c_int
    lib.ce_denoiser_destroy.argtypes = [vp]
    lib.ce_denoiser_destroy.restype = None
    lib.ce_denoise_batch.argtypes = [vp, vp, C.c_int64, C.c_int32, C.c_int32, C.c_int32, vp]
    lib.ce_denoise_batch.restype = C.c_int
    if lib.ce_abi_version() != CE_ABI_VERSION:
        raise RuntimeError(f"libce_hip.so ABI {lib.ce_abi_version()} != binding {CE_ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def last_error() -> str:
    return load().ce_last_error().decode("utf-8", "replace")
