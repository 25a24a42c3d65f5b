"""Build recipe for libgsrast.so (HIP, gfx950 only).

`python -m`-free: import and call build(), or run this file. hipcc
cross-compiles without a GPU, so this also runs in the CPU-only container;
the resulting .so travels to the GPU box inside the repo snapshot.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
LIB_DIR = PKG_DIR / "lib"
LIB_PATH = LIB_DIR / "libgsrast.so"
STAMP = LIB_DIR / "libgsrast.stamp"
ARCH = "gfx950"
SOURCES = ["api.hip", "project.hip", "isect.hip", "isect_bucket.hip", "raster_fwd.hip", "raster_bwd.hip",
           "init_depth.hip", "train_ops.hip", "ssim.hip", "knn.hip", "depthnet.hip", "pointcloud.hip", "rbf.hip"]
FLAGS = ["-O3", "-std=c++17", "-shared", "-fPIC", f"--offload-arch={ARCH}",
         "-munsafe-fp-atomics", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def _digest(srcs) -> str:
    h = hashlib.sha256()
    for p in sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.h"))
                    + [PKG_DIR.parent / "include" / "gsrast.h"]):
        h.update(p.name.encode())
        h.update(p.read_bytes())
    h.update(" ".join(FLAGS + srcs).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile csrc/*.hip into lib/libgsrast.so; no-op when up to date."""
    srcs = [s for s in SOURCES if (CSRC / s).exists()]
    LIB_DIR.mkdir(exist_ok=True)
    digest = _digest(srcs)
    if not force and LIB_PATH.exists() and STAMP.exists() and STAMP.read_text() == digest:
        return LIB_PATH
    cmd = [_hipcc(), *FLAGS, *[str(CSRC / s) for s in srcs], "-o", str(LIB_PATH)]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    STAMP.write_text(digest)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
