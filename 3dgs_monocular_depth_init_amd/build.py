"""Build recipe for libgsrast.so (HIP, gfx950 only).

`python -m`-free: import and call build(), or run this file. hipcc
cross-compiles without a GPU, so this also runs in the CPU-only container;
the resulting .so travels to the GPU box inside the repo snapshot.
"""
from __future__ import annotations

import os
import sys

# run as a script, the package directory leads sys.path and its types.py would shadow the stdlib module
sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != os.path.dirname(os.path.abspath(__file__))]

import hashlib  # noqa: E402
import shutil  # noqa: E402
import subprocess  # noqa: E402
from pathlib import Path  # noqa: E402

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


def _digest(srcs, extra=()) -> str:
    h = hashlib.sha256()
    for p in sorted(list(CSRC.glob("*.h")) + [PKG_DIR.parent / "include" / "gsrast.h"]
                    + [CSRC / s for s in srcs]):
        h.update(p.name.encode())
        h.update(p.read_bytes())
    h.update(" ".join(FLAGS + list(extra) + list(srcs)).encode())
    return h.hexdigest()


def _compile_one(args):
    hipcc, src, obj, stamp, digest, flags, verbose = args
    cmd = [hipcc, *[f for f in flags if f != "-shared"], "-c", str(src), "-o", str(obj)]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    stamp.write_text(digest)
    return obj


def build(force: bool = False, verbose: bool = False, extra_flags=(), out: Path | None = None,
          jobs: int | None = None) -> Path:
    """Compile csrc/*.hip into lib/libgsrast.so; no-op when up to date.

    Every source is its own object (lib/obj/<name>.o, rebuilt only when that source, a header or the
    flags changed; compiled `jobs` at a time), then one link. `extra_flags` / `out` build an A/B variant
    (tools/build_variants.sh) beside the product library without touching it."""
    from concurrent.futures import ThreadPoolExecutor

    srcs = [s for s in SOURCES if (CSRC / s).exists()]
    LIB_DIR.mkdir(exist_ok=True)
    out = Path(out) if out else LIB_PATH
    stamp = out.with_suffix(".stamp")
    digest = _digest(srcs, extra_flags)
    if not force and out.exists() and stamp.exists() and stamp.read_text() == digest:
        return out
    tag = hashlib.sha256(" ".join(extra_flags).encode()).hexdigest()[:8] if extra_flags else "default"
    obj_dir = LIB_DIR / "obj" / tag
    obj_dir.mkdir(parents=True, exist_ok=True)
    hipcc, flags = _hipcc(), FLAGS + list(extra_flags)
    todo, objs = [], []
    for s in srcs:
        obj, ostamp = obj_dir / (s + ".o"), obj_dir / (s + ".stamp")
        d = _digest([s], extra_flags)
        objs.append(obj)
        if force or not obj.exists() or not ostamp.exists() or ostamp.read_text() != d:
            todo.append((hipcc, CSRC / s, obj, ostamp, d, flags, verbose))
    with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(_compile_one, todo))
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *[str(o) for o in objs], "-o", str(out)]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    stamp.write_text(digest)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
