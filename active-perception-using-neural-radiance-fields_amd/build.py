"""Build libmi355nerf.so (gfx950) in-tree with hipcc.  No CPU fallback exists: if this library
is missing or has no device to run on, every compute entry point of the package raises."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libmi355nerf.so")
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


def _deps_mtime():
    m = 0.0
    for root in (CSRC, INCLUDE):
        for f in os.listdir(root):
            if f.endswith((".h", ".hip", ".cpp")):
                m = max(m, os.path.getmtime(os.path.join(root, f)))
    return m


def _compile(item):
    key, extra = item
    src = key.split("@")[0]
    obj = os.path.join(OBJ, key.replace("@", "_") + ".o")
    cmd = ["hipcc"] + COMMON + extra + os.environ.get("MNF_EXTRA_FLAGS", "").split() + ["-c", os.path.join(CSRC, src), "-o", obj]
    subprocess.check_call(cmd)
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    if os.environ.get("MNF_LIB_PATH"):          # an experiment build is in use: leave it alone
        return os.environ["MNF_LIB_PATH"]
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _deps_mtime():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    if verbose:
        print("[mi355nerf] compiling", ", ".join(SOURCES), file=sys.stderr)
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(_compile, SOURCES.items()))
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
