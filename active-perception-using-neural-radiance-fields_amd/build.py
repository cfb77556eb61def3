"""Build libmi355nerf.so (gfx950) in-tree with hipcc.  No CPU fallback exists: if this library
is missing or has no device to run on, every compute entry point of the package raises.

Two libraries come out of one set of sources:

  libmi355nerf.so        the product.  No environment variable changes what it computes or how it schedules work.
  libmi355nerf_diag.so   the same code compiled with -DMNF_DIAG: the diagnostic knobs of tools/README.md (MNF_ROUND_LOG,
                         MNF_MIN_SAMPLES, MNF_COMPOSITE_GENERAL, MNF_FIELD_SPLIT / _PIPE, MNF_HASH_BWD_*) are read from the
                         environment.  Loaded only when MNF_LIB_PATH points at it (tools/, one compositing test).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libmi355nerf.so")
LIB_DIAG = os.path.join(HERE, "libmi355nerf_diag.so")
OBJ = os.path.join(HERE, "csrc", "_obj")

COMMON = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-I" + INCLUDE, "-I" + CSRC]
# march.hip / render.hip hold the bit-exact marcher: no FMA contraction (see csrc/march_dev.h)
# value: (source file, extra flags); field / train are built twice: fp16 and bf16 matrix-core operands (csrc/common.h)
SOURCES = {
    "api.cpp": [],
    "march.hip": ["-ffp-contract=off"],
    "render.hip": ["-ffp-contract=off"],
    "field.hip": ["-fno-slp-vectorize"],   # packed-fp32 pairing of the compositing butterflies keeps their DPP operands from folding into the adds
    "train.hip": [],
    "composite_train.hip": [],
    "occupancy.hip": ["-ffp-contract=off"],
    "vanilla.hip": [],
    "trainstep.hip": ["-ffp-contract=off"],
    "field.hip@bf16": ["-DMNF_BF16", "-fno-slp-vectorize"],
    "train.hip@bf16": ["-DMNF_BF16"],
}
# translation units that read diagnostic knobs (csrc/common.h: diag_env): recompiled with -DMNF_DIAG for the diag library
DIAG_UNITS = ("render.hip", "field.hip", "train.hip", "field.hip@bf16", "train.hip@bf16", "trainstep.hip")


def _deps_mtime():
    m = 0.0
    for root in (CSRC, INCLUDE):
        for f in os.listdir(root):
            if f.endswith((".h", ".hip", ".cpp")):
                m = max(m, os.path.getmtime(os.path.join(root, f)))
    return m


def _headers_mtime():
    m = os.path.getmtime(os.path.abspath(__file__))
    for root in (CSRC, INCLUDE):
        for f in os.listdir(root):
            if f.endswith(".h"):
                m = max(m, os.path.getmtime(os.path.join(root, f)))
    return m


def _compile(item):
    key, extra, diag, force, hdr_m = item
    src = key.split("@")[0]
    obj = os.path.join(OBJ, key.replace("@", "_") + ("_diag" if diag else "") + ".o")
    src_path = os.path.join(CSRC, src)
    env_flags = os.environ.get("MNF_EXTRA_FLAGS", "").split()
    stamp = obj + ".flags"
    flags = " ".join(extra + env_flags)
    fresh = (not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src_path), hdr_m)
             and os.path.exists(stamp) and open(stamp).read() == flags)
    if not fresh:
        cmd = ["hipcc"] + COMMON + extra + (["-DMNF_DIAG"] if diag else []) + env_flags + ["-c", src_path, "-o", obj]
        subprocess.check_call(cmd)
        with open(stamp, "w") as f:
            f.write(flags)
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    if os.environ.get("MNF_LIB_PATH") and os.path.abspath(os.environ["MNF_LIB_PATH"]) not in (LIB, LIB_DIAG):
        return os.environ["MNF_LIB_PATH"]       # an experiment build is in use: leave it alone
    newest = _deps_mtime()
    if not force and all(os.path.exists(p) and os.path.getmtime(p) >= newest for p in (LIB, LIB_DIAG)):
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    if verbose:
        print("[mi355nerf] compiling", ", ".join(SOURCES), "(+ diag units)", file=sys.stderr)
    hdr_m = _headers_mtime()
    jobs = [(k, e, False, force, hdr_m) for k, e in SOURCES.items()] + [(k, SOURCES[k], True, force, hdr_m) for k in DIAG_UNITS]
    with ThreadPoolExecutor(max_workers=min(20, os.cpu_count() or 4)) as ex:
        objs = list(ex.map(_compile, jobs))
    rel = objs[:len(SOURCES)]
    diag_of = dict(zip(DIAG_UNITS, objs[len(SOURCES):]))
    dia = [diag_of.get(k, o) for k, o in zip(SOURCES, rel)]
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + rel)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_DIAG] + dia)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
