"""Build libcobevt_hip.so, libcobevt_hip_f32s.so and libcobevt_hip_f32h.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

`python -m cobevt_amd.build` or `__graft_entry__.build()`.  Objects are rebuilt only when a source or
header is newer than the object.  The second library is the SAME sources compiled with -DCOBEVT_F32_SPLIT=1
(csrc/common.hpp): its fp32-storage kernels take every matrix product through two split-bf16 MFMAs instead of
four v_mfma_f32_32x32x2_f32 - the "fp32 storage, split-bf16 matrix path" compute mode of host.set_compute_dtype.  The third
(-DCOBEVT_F32_SPLIT=2) takes them through ONE fp16 MFMA with the weight operand as a single fp16 term: the ResNet encoder's library
under host.set_compute_dtype("fp32_fast").
"""
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libcobevt_hip.so")
LIB_F32S = os.path.join(CSRC, "libcobevt_hip_f32s.so")
LIB_F32H = os.path.join(CSRC, "libcobevt_hip_f32h.so")
SOURCES = ["igemm.hip", "conv3x3.hip", "basicblock.hip", "bottleneck.hip", "bottleneck_f32.hip", "gemm_rows.hip", "gemm_rows3.hip", "gemm_rows3_f32.hip", "bev_query.hip", "row_chain.hip", "row_chain_f32.hip", "row_chain64.hip", "ln_linear64.hip", "proj_chain128.hip", "proj_chain_k.hip", "swap_stage.hip", "stem7x7.hip", "attention.hip", "attention_resident.hip", "attention_bwd.hip", "train_rows.hip", "train_glue.hip", "train_prep.hip", "wgrad3.hip", "train_nusc.hip", "train_fax.hip", "elementwise.hip", "pairwise_fusion.hip", "postprocess.hip", "depthwise.hip", "peer_gather.hip", "calibrate.hip"]
HEADERS = ["common.hpp", "attn_common.hpp", "warp_common.hpp", "row_chain.hpp", "bev_query.hpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed"]
# per-source extra flags (none at present)
EXTRA_FLAGS = {}


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _build_one(lib, objdir, extra, force, verbose):
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc] + FLAGS + extra + EXTRA_FLAGS.get(src, []) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, out.decode(errors="replace")))
    if force or _stale(lib, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return lib


def build(force=False, verbose=True):
    _build_one(LIB_F32S, os.path.join(CSRC, "f32s"), ["-DCOBEVT_F32_SPLIT=1"], force, verbose)
    _build_one(LIB_F32H, os.path.join(CSRC, "f32h"), ["-DCOBEVT_F32_SPLIT=2"], force, verbose)
    return _build_one(LIB, CSRC, [], force, verbose)


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
