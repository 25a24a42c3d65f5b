"""Records what the parity checks actually measured (flip counts, max / mean errors, PSNR
deltas, gradient errors) so that the slack inside the tolerances is visible: the GPU run
writes gpurun_out/parity_stats.json, which is committed as profiles/parity_r<NN>.json."""
from __future__ import annotations

import json
import math
import os
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
RECORDS: list = []


def current_test() -> str:
    return os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]


def record(kind: str, **stats) -> None:
    RECORDS.append({"test": current_test(), "kind": kind, **stats})


def psnr(a, b, peak: float = 1.0) -> float:
    mse = float(((a.double() - b.double()) ** 2).mean())
    return float("inf") if mse == 0.0 else 10.0 * math.log10(peak * peak / mse)


def dump() -> None:
    if not RECORDS:
        return
    out = ROOT / "gpurun_out"
    out.mkdir(exist_ok=True)
    (out / "parity_stats.json").write_text(json.dumps(RECORDS, indent=1))
