"""``oracle/_ref``: the recipe that puts the unmodified reference on the bench box for the
``cpu_baseline`` leg (oracle/make_ref.py, oracle/ref_runner.py).  CPU only; the copy itself needs
/root/reference (build container) -- elsewhere the prebuilt copy is used if it shipped."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_make_ref_copies_the_reference_and_git_ignores_it(tmp_path):
    from oracle.make_ref import make_ref
    if not os.path.isdir("/root/reference/rlpyt"):
        pytest.skip("no /root/reference here (GPU box): the shipped oracle/_ref is used")
    dest = make_ref(dest=str(tmp_path / "_ref"), quiet=True)
    assert os.path.isfile(os.path.join(dest, "rlpyt", "algos", "pg", "ppo.py"))
    assert os.path.isfile(os.path.join(dest, "rlpyt", "samplers", "serial", "sampler.py"))
    assert not os.path.isdir(os.path.join(dest, "rlpyt", "ul"))
    # byte-identical copies: the baseline is the UNMODIFIED reference
    with open("/root/reference/rlpyt/algos/pg/ppo.py", "rb") as a, \
            open(os.path.join(dest, "rlpyt", "algos", "pg", "ppo.py"), "rb") as b:
        assert a.read() == b.read()
    # never part of the history
    out = subprocess.run(["git", "check-ignore", "oracle/_ref/rlpyt/__init__.py"], cwd=ROOT,
                         capture_output=True, text=True)
    assert out.returncode == 0, "oracle/_ref must be listed in .gitignore"
    ign = os.path.join(ROOT, ".gpurunignore")
    if os.path.exists(ign):
        assert "oracle/_ref" not in open(ign).read(), "oracle/_ref must travel to the GPU box"


def test_reference_ppo_iteration_runs_from_oracle_ref():
    """One tiny iteration of the reference's SerialSampler + PPO + AtariFfAgent out of oracle/_ref
    (in a subprocess: the reference installs its own logger state)."""
    from oracle import ref_runner
    if not ref_runner.available():
        pytest.skip("oracle/_ref not built (python oracle/make_ref.py)")
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from oracle import ref_runner as R\n"
            "from rlpyt_amd.envs.synthetic import SyntheticPong\n"
            "assert R.load()\n"
            "r = R.time_ppo(SyntheticPong, {}, T=4, B=4, iters=1, threads=2)\n"
            "assert r['value'] > 0 and r['last_loss'] == r['last_loss']\n"
            "import rlpyt; print('REF', rlpyt.__file__)\n") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert os.path.join("oracle", "_ref", "rlpyt") in out.stdout
