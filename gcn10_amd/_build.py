"""Builds the native libraries in-tree (``make`` at the repo root)."""
from __future__ import annotations

import os
import subprocess


def repo_root() -> str:
    return os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_all(targets=("gpu", "host", "cli", "oracle"), verbose: bool = False) -> None:
    """Compile every native target for gfx950 (hipcc cross-compiles without a GPU)."""
    env = dict(os.environ)
    env.setdefault("HIPCC", "/opt/rocm/bin/hipcc")
    proc = subprocess.run(["make", "-C", repo_root(), *targets], env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or proc.returncode != 0:
        print(proc.stdout)
    if proc.returncode != 0:
        raise RuntimeError("native build failed (make %s)" % " ".join(targets))
