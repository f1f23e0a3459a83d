"""Builds libgnnrag_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m gnnrag_amd.build        (or: __graft_entry__.build())

One object per .hip file (compiled in parallel), linked into gnn-rag_amd/lib/libgnnrag_hip.so.
The library depends on the HIP runtime only - no torch, no Python.
"""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libgnnrag_hip.so")
ARCH = "gfx950"
SOURCES = ["csr_plan.hip", "aggregate.hip", "aggregate_bwd.hip", "gemm_f32.hip", "tables_b3.hip", "softmax_layer.hip", "rel_transform.hip", "gemm_tn.hip", "eval_tail.hip", "query_update.hip", "frontier.hip", "lstm.hip"]
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(REPO, "include"),
         "-I" + CSRC, "-Wno-unused-result"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest(paths) -> str:
    h = hashlib.sha256()
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(p.encode())
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _deps():
    d = [os.path.join(CSRC, s) for s in SOURCES]
    d += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    d.append(os.path.join(REPO, "include", "gnnrag.h"))
    return d


def build_variant(name: str, defines: dict, verbose: bool = False) -> str:
    """Experiment builds: lib/exp_<name>.so compiled with extra -D switches (tools/tune_variants.py)."""
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    objdir = os.path.join(LIBDIR, "obj_" + name)
    os.makedirs(objdir, exist_ok=True)
    # "__flags__": extra compiler flags of the variant (e.g. ["-fno-slp-vectorize"]); everything else is a -D switch
    extra = list(defines.get("__flags__", [])) + ["-D%s=%s" % kv for kv in defines.items() if not kv[0].startswith("__")]

    def cc(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        r = subprocess.run([hipcc] + FLAGS + extra + ["-c", os.path.join(CSRC, src), "-o", obj],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s (%s):\n%s" % (src, name, r.stderr[-4000:]))
        return obj

    with cf.ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(cc, SOURCES))
    out = os.path.join(LIBDIR, "exp_%s.so" % name)
    r = subprocess.run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-pthread", "-o", out] + objs,
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed (%s):\n%s" % (name, r.stderr[-4000:]))
    if verbose:
        print("built", out)
    return out


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, "build.sha256")
    dig = _digest(_deps())
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)

    def cc(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, " ".join(cmd), r.stderr[-8000:]))
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        return obj

    with cf.ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(cc, SOURCES))
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-pthread", "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-8000:]))
    with open(stamp, "w") as f:
        f.write(dig)
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
