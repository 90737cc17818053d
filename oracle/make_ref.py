"""Recipe for ``oracle/_ref/``: put the UNMODIFIED reference on the path of the bench box.

TEST / BASELINE INFRASTRUCTURE ONLY (nothing under ``rlpyt_amd/`` may import it).

The reference is pure Python, so "building" it is copying the package where it lies under
``/root/reference`` (only present in the build container) into ``oracle/_ref/rlpyt`` -- which is
git-ignored (no reference source ever enters the history) but NOT gpurun-ignored, so it travels to
the GPU box with the snapshot exactly like the built ``.so`` files.  ``__graft_entry__.build()``
runs this when ``/root/reference`` exists; ``oracle/ref_runner.py`` imports the copy for
``bench.py``'s ``cpu_baseline`` leg (``kind: "reference"``) and falls back to the CPU port
(``kind: "port"``) when the copy is absent.

Copied: every ``.py`` of ``rlpyt/`` that the hot path's CPU run can reach (samplers, collectors,
agents, algos, models, replays, distributions, spaces, utils, envs/base).  Left out: ``ul/``,
``projects/``, ``experiments/`` (not on the path, SURVEY section 2 "out of scope").
"""
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SKIP_DIRS = {"ul", "projects", "experiments", "__pycache__"}


def make_ref(reference_root="/root/reference", dest=None, quiet=False):
    """Copy the reference package; returns the destination or None when there is no reference."""
    src = os.path.join(reference_root, "rlpyt")
    dest = dest or os.path.join(HERE, "_ref")
    if not os.path.isdir(src):
        return None
    out = os.path.join(dest, "rlpyt")
    if os.path.isdir(out):
        shutil.rmtree(out)
    n = 0
    for root, dirs, files in os.walk(src):
        dirs[:] = [d for d in dirs if d not in SKIP_DIRS]
        rel = os.path.relpath(root, src)
        os.makedirs(os.path.join(out, rel), exist_ok=True)
        for f in files:
            if f.endswith(".py"):
                shutil.copy2(os.path.join(root, f), os.path.join(out, rel, f))
                n += 1
    lic = os.path.join(reference_root, "LICENSE")
    if os.path.exists(lic):
        shutil.copy2(lic, os.path.join(dest, "LICENSE.rlpyt"))
    with open(os.path.join(dest, "README"), "w") as f:
        f.write("Unmodified copy of astooke/rlpyt's python package (made by oracle/make_ref.py from "
                f"{src}).\nGit-ignored on purpose: baseline infrastructure, not product source.\n")
    if not quiet:
        print(f"oracle/_ref: {n} reference files copied from {src}")
    return dest


if __name__ == "__main__":
    d = make_ref(*(sys.argv[1:2] or ["/root/reference"]))
    sys.exit(0 if d else 1)
